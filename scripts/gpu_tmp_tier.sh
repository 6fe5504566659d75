#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 900 python -X faulthandler -m pytest tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/u_test_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/u_test_$i.log
  tail -2 gpurun_out/u_test_$i.log | tr '\n' ' '; echo
done
