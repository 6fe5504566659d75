#!/bin/bash
# round-2 GPU session C: fused score kernels + faster chain; engine tier; bench + profile
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "score_fused or gru_chain" 2>&1 | tail -30) > gpurun_out/c_kernels.log
(timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py -q -rP -p no:cacheprovider 2>&1) > gpurun_out/c_engine_full.log
grep -E "passed|failed|error|bf16 anchor:|cfg5 fused|Error|assert" gpurun_out/c_engine_full.log | tail -40 > gpurun_out/c_engine.log
(timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/c_bench_cfg2.log
(timeout 300 python bench.py --config cfg5 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/c_bench_cfg5.log
(timeout 300 python bench.py --config cfg4 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/c_bench_cfg4.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02c -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/c_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r02c/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r02c_stats.txt 2>&1 && python scripts/timeline.py $f "gru|score|igemm_kernel|ce_|colsum|wgrad2|reduce_unpack|copy2d|transpose|tpool|adam" > gpurun_out/prof_r02c_timeline.txt 2>&1
rm -rf gpurun_out/prof_r02c/*.db
ls gpurun_out | head -60
