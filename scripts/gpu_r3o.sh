#!/bin/bash
# round 3, session o: strided input-gradient variants (class rotation, lighter scalar code) + phase probe of the PAR kernel
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dgrad" 2>&1 | tail -3) > gpurun_out/o_test_dgrad.log
S="l2s l3s l4s l2s224 l3s224"
(echo "# default"; timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_PARITY_ILV=0"; DPC_PARITY_ILV=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_IGEMM_WS_PAR=0"; DPC_IGEMM_WS_PAR=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 (round-3 start)"; DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F) > gpurun_out/o_conv_bench.txt
(for sh in l3s l4s; do echo "# $sh"; WS_PROBE_SHAPE=$sh timeout 200 python scripts/probes/ws_probe.py 2>&1 | $F; done
 echo "# l3s DPC_PARITY_ILV=0"; DPC_PARITY_ILV=0 WS_PROBE_SHAPE=l3s timeout 200 python scripts/probes/ws_probe.py 2>&1 | $F) > gpurun_out/o_ws_probe.txt
cat gpurun_out/o_test_dgrad.log gpurun_out/o_conv_bench.txt gpurun_out/o_ws_probe.txt
