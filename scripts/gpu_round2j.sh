#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv or gemm or gru or score" 2>&1 | tail -3) > gpurun_out/j_tests.log
(timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3) >> gpurun_out/j_tests.log
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/j_bench_cfg2.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02j -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/j_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r02j/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r02j_stats.txt 2>&1
rm -rf gpurun_out/prof_r02j/*.db
cat gpurun_out/j_tests.log; for f in j_bench_cfg2; do python -c "import json,sys; d=json.loads(open('gpurun_out/$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'])"; done; grep -E "igemm" gpurun_out/prof_r02j_stats.txt
