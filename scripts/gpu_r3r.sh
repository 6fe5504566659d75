#!/bin/bash
# round 3, session r: stem max-pool forward with all window loads in flight
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "pool or stem" 2>&1 | tail -3) > gpurun_out/r_test.log
(timeout 200 python scripts/stem_bench.py 2>&1 | $F) > gpurun_out/r_stem_bench.txt
B="python bench.py --no-cpu-baseline --no-also --steps 100"
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/r_bench_cfg2.log
cat gpurun_out/r_test.log gpurun_out/r_stem_bench.txt
python - <<'P'
import json
d=json.loads(open('gpurun_out/r_bench_cfg2.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['hbm_family'])
P
