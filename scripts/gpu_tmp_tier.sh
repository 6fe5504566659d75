#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python -X faulthandler -m pytest tests/test_block_grads_gpu.py tests/test_checkpoint_gpu.py tests/test_configs_gpu.py tests/test_cotenant_gpu.py tests/test_data_pipeline.py tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py -m gpu -v -p no:cacheprovider --durations=5 > gpurun_out/s_test_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/s_test_$i.log
  tail -4 gpurun_out/s_test_$i.log
done
dmesg 2>/dev/null | tail -5
