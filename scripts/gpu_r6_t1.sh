#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest "tests/test_engine_gpu.py::test_bf16_anchored_to_reference" -m gpu -q -p no:cacheprovider -s 2>&1 | tail -170) > gpurun_out/r06_anchor34.log
(timeout 1500 python -m pytest tests/test_block_grads_gpu.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "backbone|passed|failed|Error|assert" | tail -40) > gpurun_out/r06_blockgrads.log
cat gpurun_out/r06_anchor34.log | grep -E "anchor|assert|Error|passed|failed|^E" | head -40; cat gpurun_out/r06_blockgrads.log
