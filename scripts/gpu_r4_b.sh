#!/bin/bash
# round 4, session B: squatter repro with real co-residency (one 4-wave workgroup per CU), round 3's own repro (corun_probe: weight
# gradient + slab reduce on a side stream) against the nofix and the fixed library, the new split-K GEMMs, head timings, bench.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm" 2>&1 | tail -5) > gpurun_out/b_test_gemm.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/b_head.log
(timeout 300 python scripts/probes/squat_probe.py 1000 l3 2>&1 | grep -v amdgpu.ids) > gpurun_out/b_squat_l3.log
(timeout 200 python scripts/probes/squat_probe.py 400 l2 2>&1 | grep -v amdgpu.ids) > gpurun_out/b_squat_l2.log
(DPC_PROBE_LIB=scripts/probes/libdpc_nofix.so timeout 300 python scripts/probes/corun_probe.py l3 15000 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/b_corun_nofix.log
(timeout 300 python scripts/probes/corun_probe.py l3 15000 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/b_corun_fixed.log
(timeout 400 python -m pytest tests/test_cotenant_gpu.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/b_test_engine.log
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/b_bench_cfg2.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1) > gpurun_out/b_bench_torchrun.log
cat gpurun_out/b_test_gemm.log gpurun_out/b_head.log gpurun_out/b_squat_l3.log gpurun_out/b_squat_l2.log gpurun_out/b_corun_nofix.log gpurun_out/b_corun_fixed.log gpurun_out/b_test_engine.log
python - <<'P'
import json
for f in ("b_bench_cfg2","b_bench_torchrun"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.log").read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('score_gemm'), d.get('schedules'), {k:v.get('value') for k,v in d.get('also',{}).items()})
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{f}.log").read()[-600:])
P
