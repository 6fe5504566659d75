#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(for d in 0 1 2 4 36 8 24 12 60 63 127; do echo "DPC_SF_DBG=$d"; DPC_SF_DBG=$d timeout 100 python scripts/stem_bench.py --probe --iters 5 2>&1 | grep -v amdgpu.ids | grep fused; done) > gpurun_out/g_stem_probe.txt
cat gpurun_out/g_stem_probe.txt
