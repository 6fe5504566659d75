#!/bin/bash
# round 4: workgroup caps of the unrolled BatchNorm apply kernels (DPC_BN_APPLY_GRID forward, DPC_BN_BWD_GRID backward) in the step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for gg in "8192 8192" "16384 16384" "8192 16384" "16384 8192" "8192 32768" "8192 8192" "16384 16384"; do set -- $gg; (DPC_BN_APPLY_GRID=$1 DPC_BN_BWD_GRID=$2 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('apply=$1 bwd=$2', d['value'], d['ms_per_step'], 'hbm', d['hbm_family']['ms_per_step'], d['hbm_family']['achieved'])"); done > gpurun_out/i_bench2.log
cat gpurun_out/i_bench2.log
