#!/bin/bash
# round 5 closing session: GPU tier, smoke, the default bench line (live PMC, also.*), bench under torch.distributed.run (1 rank, with the side
# schedules), rocprofv3 kernel stats of cfg2 / cfg4 / cfg5 + cfg2 timeline, PMC traffic per kernel, head kernel timings
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/z_test_full.log 2>&1; echo "rc=$?" >> gpurun_out/z_test_full.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/z_smoke.log
(timeout 900 python bench.py 2>gpurun_out/z_bench.err | tail -1) > gpurun_out/z_bench_cfg2.log
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 3 --no-cpu-baseline --no-roofline --schedules 2>&1 | tail -1) > gpurun_out/z_bench_torchrun.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/z_head.log
bash scripts/gpu_r5_prof.sh > /dev/null 2>&1
PMC_TAG=r05 bash scripts/gpu_pmc_traffic.sh > gpurun_out/z_pmc_traffic.log 2>&1
cp profiles/r05_pmc_traffic.json profiles/r05_pmc_traffic_per_kernel.txt gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
tail -4 gpurun_out/z_test_full.log; cat gpurun_out/z_smoke.log
python - <<'P'
import json
for f in ("z_bench_cfg2", "z_bench_torchrun"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.log").read().strip().splitlines()[-1])
        r=d.get('roofline',{})
        print(f, d['value'], d['ms_per_step'], "conv", r.get('frac'), (r.get('executed') or {}).get('frac'), "traffic", r.get('traffic'), (r.get('traffic_source') or {}).get('live'), "score", d.get('score_gemm',{}).get('frac'), d.get('score_gemm',{}).get('us_per_step'), "hbm", d.get('hbm_family',{}).get('ms_per_step'), d.get('schedules'))
        for k,v in d.get('also',{}).items(): print("   ", k, {a:b for a,b in v.items() if a in ('value','ms_per_step','error','vs_engine_path','two_stream_speedup','score_path')}, (v.get('score_gemm') or {}).get('frac'))
        if 'cpu_baseline' in d: print("   cpu", d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.log").read()[-300:])
P
