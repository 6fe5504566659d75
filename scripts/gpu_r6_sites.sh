#!/bin/bash
# round 6, session 1: which weight gradients belong on the side stream (scripts/probes/side_sites.py), cfg2 / cfg4 / cfg5; the new GPU tests
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python scripts/probes/side_sites.py cfg2 30 2>&1 | tail -80) > gpurun_out/r06_side_sites_cfg2.txt
(timeout 500 python scripts/probes/side_sites.py cfg4 12 2>&1 | tail -120) > gpurun_out/r06_side_sites_cfg4.txt
(timeout 600 python -m pytest tests/test_graph_rccl_gpu.py tests/test_data_pipeline.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15) > gpurun_out/r06_s1_tests.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/r06_s1_bench.log
cat gpurun_out/r06_s1_tests.log; head -c 1500 gpurun_out/r06_s1_bench.log; echo; cat gpurun_out/r06_side_sites_cfg2.txt
