#!/bin/bash
# round 4 full session: GPU tier, smoke, the default bench line (also: cfg4, cfg5, module, f32, f32_bf16x6; CPU baseline), one-stream
# and torchrun lines, head kernel timings, rocprofv3 kernel trace (stats + timeline) of cfg2 / cfg4 / cfg5, PMC traffic
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > gpurun_out/f_test_full.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/f_smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/f_bench_cfg2.log
(DPC_WGRAD_STREAM=0 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1) > gpurun_out/f_bench_cfg2_one_stream.log
(timeout 300 python bench.py --config cfg5 --score-path materialised --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/f_bench_cfg5_mat.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1) > gpurun_out/f_bench_torchrun.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/f_head.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/f_rocprof.log
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f4 -o bench -- python $R/bench.py --config cfg4 --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/f_rocprof4.log
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f5 -o bench -- python $R/bench.py --config cfg5 --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/f_rocprof5.log
cd $R
f=$(ls gpurun_out/prof_f/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f 4 > gpurun_out/f_kernel_stats_cfg2.txt 2>&1 && python scripts/timeline.py $f "." > gpurun_out/f_timeline_cfg2.txt 2>&1
f=$(ls gpurun_out/prof_f4/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f 3 > gpurun_out/f_kernel_stats_cfg4.txt 2>&1
f=$(ls gpurun_out/prof_f5/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f 3 > gpurun_out/f_kernel_stats_cfg5.txt 2>&1
rm -rf gpurun_out/prof_f gpurun_out/prof_f4 gpurun_out/prof_f5
PMC_TAG=r04 bash scripts/gpu_pmc_traffic.sh > gpurun_out/f_pmc_traffic.log 2>&1
cp profiles/r04_pmc_traffic.json profiles/r04_pmc_traffic_per_kernel.txt gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/f_test_full.log gpurun_out/f_smoke.log
python - <<'P'
import json
for f in ("f_bench_cfg2","f_bench_cfg2_one_stream","f_bench_cfg5_mat","f_bench_torchrun"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.log").read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], "conv", d.get('roofline',{}).get('frac'), "score", d.get('score_gemm',{}).get('frac'), d.get('score_gemm',{}).get('us_per_step'), "hbm", d.get('hbm_family',{}).get('ms_per_step'), d.get('schedules'))
        for k,v in d.get('also',{}).items(): print("   ", k, {a:b for a,b in v.items() if a in ('value','ms_per_step','error','vs_engine_path','dtype')})
        if 'cpu_baseline' in d: print("   cpu", d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.log").read()[-300:])
P
ls gpurun_out | tr '\n' ' '
