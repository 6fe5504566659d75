#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "stem" 2>&1 | tail -3) > gpurun_out/h_tests.log
(timeout 300 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1) > gpurun_out/h_bench_cfg2.log
(DPC_STEM_FUSED=0 timeout 300 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1) > gpurun_out/h_bench_cfg2_unfused.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02h -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/h_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r02h/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r02h_stats.txt 2>&1
rm -rf gpurun_out/prof_r02h/*.db
cat gpurun_out/h_tests.log; for f in h_bench_cfg2 h_bench_cfg2_unfused; do python -c "import json,sys; d=json.loads(open('gpurun_out/$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['wgrad_kernel'])"; done; grep -E "wgrad_stem|pool_bn" gpurun_out/prof_r02h_stats.txt
