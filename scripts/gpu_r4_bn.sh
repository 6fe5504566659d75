#!/bin/bash
# round 4: units in flight per thread in the streaming BatchNorm kernels (DPC_BN_UNROLL), layer1 size of cfg2 and the step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for u in 1 2 4; do echo "== DPC_BN_UNROLL=$u"; DPC_BN_UNROLL=$u timeout 120 python scripts/elt_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/h_elt.log
for g in 2048 4096; do echo "== DPC_BN_UNROLL=4 DPC_BN_APPLY_GRID=$g"; DPC_BN_UNROLL=4 DPC_BN_APPLY_GRID=$g timeout 120 python scripts/elt_bench.py 2>&1 | grep "bn_apply"; done >> gpurun_out/h_elt.log
for u in 1 4 1 4; do (DPC_BN_UNROLL=$u timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DPC_BN_UNROLL=$u', d['value'], d['ms_per_step'], 'hbm', d['hbm_family']['ms_per_step'], d['hbm_family']['achieved'])"); done > gpurun_out/h_bench.log
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "bn" 2>&1 | tail -3) > gpurun_out/h_test.log
cat gpurun_out/h_elt.log gpurun_out/h_bench.log gpurun_out/h_test.log
