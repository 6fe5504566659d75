#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -2) > gpurun_out/e_tests.log
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/e_bench_cfg2.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02e -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/e_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r02e/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r02e_stats.txt 2>&1 && python scripts/timeline.py $f "reduce_unpack|pack3d" > gpurun_out/prof_r02e_timeline.txt 2>&1
rm -rf gpurun_out/prof_r02e/*.db
cat gpurun_out/e_tests.log; cut -c1-400 gpurun_out/e_bench_cfg2.log; grep -E "reduce_unpack|pack3d" gpurun_out/prof_r02e_stats.txt; cat gpurun_out/prof_r02e_timeline.txt | head -40
