#!/bin/bash
# the whole GPU tier + smoke on the current tree
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25) > gpurun_out/r06_gpu_tier.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) >> gpurun_out/r06_gpu_tier.txt
cat gpurun_out/r06_gpu_tier.txt
