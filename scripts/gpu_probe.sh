#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out

(timeout 300 python scripts/probes/ws_probe.py 2>&1 | tail -20) > gpurun_out/ws_probe.txt
cat gpurun_out/ws_probe_tests.txt gpurun_out/ws_probe.txt
