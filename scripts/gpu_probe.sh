#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -3) > gpurun_out/wgrad_probe.txt
(timeout 300 python scripts/probes/wgrad_probe.py 2>&1 | tail -24) >> gpurun_out/wgrad_probe.txt
cat gpurun_out/wgrad_probe.txt
