#!/bin/bash
# round 4: the ConvGRU recurrence on eight waves per workgroup (one column tile per wave) against four (DPC_GRU_WAVES)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for w in 4 8; do echo "== DPC_GRU_WAVES=$w"; DPC_GRU_WAVES=$w timeout 300 python scripts/head_bench.py 2>&1 | grep "gru_chain"; done > gpurun_out/q_gru.log
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_lc_gpu.py -m gpu -q -p no:cacheprovider -k "gru or lc" 2>&1 | tail -3) > gpurun_out/q_test.log
for w in 4 8 4 8; do (DPC_GRU_WAVES=$w timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DPC_GRU_WAVES=$w', d['value'], d['ms_per_step'])"); done > gpurun_out/q_bench.log
cat gpurun_out/q_gru.log gpurun_out/q_test.log gpurun_out/q_bench.log
