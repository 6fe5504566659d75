#!/bin/bash
# round 4, session C: full GPU tier on the current build, the default bench line (with also.module / also.f32 and the CPU baseline),
# head kernel timings, rocprofv3 kernel trace (stats + timeline of one step)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25) > gpurun_out/c_test_full.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/c_smoke.log
(timeout 600 python bench.py 2>&1 | tail -1) > gpurun_out/c_bench_cfg2.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/c_head.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/c_rocprof.log
cd $R
f=$(ls gpurun_out/prof_c/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f 4 > gpurun_out/c_kernel_stats.txt 2>&1
[ -n "$f" ] && python scripts/timeline.py $f "." > gpurun_out/c_timeline.txt 2>&1
rm -rf gpurun_out/prof_c
cat gpurun_out/c_test_full.log gpurun_out/c_smoke.log gpurun_out/c_head.log
head -40 gpurun_out/c_kernel_stats.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/c_bench_cfg2.log").read().strip().splitlines()[-1])
print("cfg2", d['value'], d['ms_per_step'], "conv frac", d.get('roofline',{}).get('frac'), "score", d.get('score_gemm',{}).get('frac'), d.get('score_gemm',{}).get('us_per_step'), "hbm", d.get('hbm_family',{}).get('ms_per_step'))
for k,v in d.get('also',{}).items(): print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','error','vs_engine_path','final_loss')})
print(d.get('cpu_baseline'))
P
