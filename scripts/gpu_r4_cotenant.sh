#!/bin/bash
# round 4: the co-tenant experiments behind profiles/r04_cotenant.txt (sessions A and B of the round).
#   1. squatter on every CU while the layer3 / layer2 input-gradient runs: round 3's code (make nofix) against the fixed library
#   2. round 3's own repro (weight gradient + slab reduction on a side stream, 15 000 launches), both libraries
#   3. what a co-tenant that owns k CUs costs the train step under the three exchange schedules
#   4. the GPU tier beside a co-tenant
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 300 python scripts/probes/squat_probe.py 1000 l3 2>&1 | grep -v amdgpu.ids) > gpurun_out/co_squat_l3.log
(timeout 200 python scripts/probes/squat_probe.py 400 l2 2>&1 | grep -v amdgpu.ids) > gpurun_out/co_squat_l2.log
(DPC_PROBE_LIB=scripts/probes/libdpc_nofix.so timeout 300 python scripts/probes/corun_probe.py l3 15000 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/co_corun_nofix.log
(timeout 300 python scripts/probes/corun_probe.py l3 15000 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/co_corun_fixed.log
(timeout 400 python scripts/probes/cotenant_step.py cfg2 1000 20 8,16,32 2>&1 | grep -v amdgpu.ids) > gpurun_out/co_cotenant_step.log
(timeout 600 python -m pytest tests/test_cotenant_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/co_test_cotenant.log
cat gpurun_out/co_*.log
