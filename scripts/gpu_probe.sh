#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 300 python scripts/probes/halo_probe.py 2>&1 | tail -12) > gpurun_out/halo_probe.txt
cat gpurun_out/halo_probe.txt
