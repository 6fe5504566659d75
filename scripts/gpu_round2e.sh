#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_lc_gpu.py tests/test_data_pipeline.py tests/test_kernels_gpu.py -m gpu -q -rP -p no:cacheprovider -k "lc or frames or wgrad or score_fused or engine_consumes" 2>&1 | grep -E "passed|failed|LC gradients|Error|assert " | tail -15) > gpurun_out/e_tests.log
(timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3) >> gpurun_out/e_tests.log
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/e_bench_cfg2.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02e -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/e_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r02e/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r02e_stats.txt 2>&1
rm -rf gpurun_out/prof_r02e/*.db
cat gpurun_out/e_tests.log; cut -c1-400 gpurun_out/e_bench_cfg2.log; grep -E "reduce_unpack|pack3d" gpurun_out/prof_r02e_stats.txt
