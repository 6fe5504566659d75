#!/bin/bash
# all phase probes with the current build ('make probe' first) -> gpurun_out/probe_*.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for p in ws wgrad halo overlap; do
  (timeout 300 python scripts/probes/${p}_probe.py 2>&1 | grep -v amdgpu.ids | tail -24) > gpurun_out/probe_$p.txt
done
tail -5 gpurun_out/probe_ws.txt
