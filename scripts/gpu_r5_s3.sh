#!/bin/bash
# round 5, session 3: the default bench line with its new legs (live PMC traffic, paired one-stream A/B, score_gemm of cfg4 / cfg5, cfg5 fused beside it)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(time timeout 900 python bench.py 2> gpurun_out/s3_bench.err | tail -1 > gpurun_out/s3_bench_cfg2.log) 2>&1 | tail -3
tail -5 gpurun_out/s3_bench.err | grep -v amdgpu.ids
python - <<'P'
import json
d=json.loads(open("gpurun_out/s3_bench_cfg2.log").read().strip().splitlines()[-1])
r=d['roofline']
print("cfg2", d['value'], d['ms_per_step'], "conv", r['frac'], "executed", r.get('executed'), "traffic", r['traffic'], r['traffic_source'])
print("score", d['score_gemm']['frac'], d['score_gemm']['us_per_step'], "hbm", d['hbm_family']['ms_per_step'], d['hbm_family']['frac'])
for k,v in d.get('also',{}).items(): print("   ", k, {a:b for a,b in v.items() if a in ('value','ms_per_step','error','vs_engine_path','two_stream_speedup','two_stream_ms_per_step','all_ms_per_step','score_path')}, v.get('score_gemm',{}).get('frac'), v.get('score_gemm',{}).get('us_per_step'), v.get('roofline',{}).get('executed_frac'))
print("cpu", d.get('cpu_baseline',{}).get('value'))
P
