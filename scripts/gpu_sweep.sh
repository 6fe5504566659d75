#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for v in 1536 1024 768 512; do
  echo "DPC_WGRAD_BLOCKS=$v $(DPC_WGRAD_BLOCKS=$v timeout 300 python bench.py --no-cpu-baseline --steps 100 --roofline-steps 2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["wgrad_kernel"])')"
done 2>&1 | tee gpurun_out/sweep_wgrad_blocks.log
for v in 256 512 1024; do
  echo "DPC_WGRAD_PATCH_BLOCKS=$v $(DPC_WGRAD_PATCH_BLOCKS=$v timeout 300 python bench.py --no-cpu-baseline --steps 100 --roofline-steps 2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["wgrad_kernel"])')"
done 2>&1 | tee gpurun_out/sweep_wgrad_patch_blocks.log
