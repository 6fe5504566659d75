#!/bin/bash
# round 4, session A: the squatter repro of round 3's "LDS race" (nofix vs fixed library), the co-tenancy cost table, the GPU tier
# on the build without the LDS claim, one bench line.  Outputs: gpurun_out/a_*.log
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 300 python scripts/probes/squat_probe.py 400 l3 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_squat_l3.log
(timeout 300 python scripts/probes/squat_probe.py 300 l2 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_squat_l2.log
(timeout 600 python -m pytest tests/test_cotenant_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/a_test_cotenant.log
(timeout 400 python scripts/probes/cotenant_step.py cfg2 1000 20 8,16,32 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_cotenant_step.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/a_bench_cfg2.log
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_cotenant_gpu.py 2>&1 | tail -15) > gpurun_out/a_test_full.log
cat gpurun_out/a_squat_l3.log gpurun_out/a_squat_l2.log gpurun_out/a_test_cotenant.log gpurun_out/a_cotenant_step.log gpurun_out/a_test_full.log
python -c "import json; d=json.loads(open('gpurun_out/a_bench_cfg2.log').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('score_gemm',{}).get('frac'))"
