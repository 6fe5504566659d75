#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "dgrad" 2>&1 | tail -5) > gpurun_out/r06_wsd_tests.log
(timeout 900 python -m pytest tests/test_block_grads_gpu.py -m gpu -q -p no:cacheprovider -k "oracle and not r34" 2>&1 | tail -3) >> gpurun_out/r06_wsd_tests.log
cat gpurun_out/r06_wsd_tests.log
STEPS=100 bash scripts/gpu_ab.sh r06_wsd_ab.txt DPC_IGEMM_WSD=0 DPC_X=1 DPC_WSD_A_PERMILLE=520 DPC_WSD_A_PERMILLE=600 DPC_WSD_A_PERMILLE=640
