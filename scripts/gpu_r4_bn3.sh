#!/bin/bash
# round 4: BatchNorm kernels on the tensors below the streaming threshold (layer3 / layer4 of cfg2) in the step: units per thread in flight, forward workgroup cap
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for k in "1 8192" "4 512" "4 256" "4 1024" "1 8192" "4 512"; do set -- $k; (DPC_BN_SMALL_UNROLL=$1 DPC_BN_SMALL_GRID=$2 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('small unroll=$1 grid(fwd+bwd)=$2', d['value'], d['ms_per_step'], 'hbm', d['hbm_family']['ms_per_step'], d['hbm_family']['achieved'])"); done > gpurun_out/k_bench2.log
(timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defaults (unroll 4, forward cap 512, backward cap 2048)', d['value'], d['ms_per_step'], 'hbm', d['hbm_family']['ms_per_step'], d['hbm_family']['achieved'])") >> gpurun_out/k_bench2.log
cat gpurun_out/k_bench2.log
