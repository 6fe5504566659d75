#!/bin/bash
# round 4: bit-identity stress of the two-stream schedule WITHOUT the 160 KB LDS claim (round 3 needed it for this run to stay clean):
# two engines, one per schedule, parameters and gradients compared bit for bit after every step (scripts/stream_stress.py); plus the
# schedule round 3 found NOT reproducible (side work beside input-gradients, DPC_SIDE_QUIET=0) -- the hazard fix is what that needed
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python scripts/stream_stress.py cfg2 1500 1 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/s_stress_cfg2.log
(timeout 600 python scripts/stream_stress.py cfg4 400 1 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/s_stress_cfg4.log
(timeout 600 python scripts/stream_stress.py cfg5 250 1 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/s_stress_cfg5.log
# (switch removed in round 5) (DPC_SIDE_QUIET=0 timeout 900 python scripts/stream_stress.py cfg2 2500 1 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/s_stress_cfg2_freerun.log
# (switch removed in round 5) (DPC_SIDE_QUIET=0 STRESS_EAGER=1 timeout 900 python scripts/stream_stress.py cfg2 1000 1 2>&1 | grep -v amdgpu.ids | tail -4) > gpurun_out/s_stress_cfg2_freerun_eager.log
for f in gpurun_out/s_stress_*.log; do echo "== $f"; cat $f; done
