#!/bin/bash
# quick check after a kernel change: kernel tests (KEXPR), config tests, one cfg2 bench line, kernel stats of 4 steps
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "${KEXPR:-conv or gemm or wgrad}" 2>&1 | tail -3) > gpurun_out/q_tests.log
(timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3) >> gpurun_out/q_tests.log
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/q_bench_cfg2.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/q_rocprof.log
cd $R
f=$(ls gpurun_out/prof_q/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_q_stats.txt 2>&1
rm -rf gpurun_out/prof_q/*.db
cat gpurun_out/q_tests.log; python -c "import json,sys; d=json.loads(open('gpurun_out/q_bench_cfg2.log').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['wgrad_kernel'])"; grep -E "${KGREP:-igemm|wgrad}" gpurun_out/prof_q_stats.txt
