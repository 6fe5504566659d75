#!/bin/bash
# round-3 session C: kernel tests of what changed, A/B of the rebuilt fused stem weight gradient, kernel stats
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -k "stem or pool or dgrad or gemm or fused" 2>&1 | tail -15) > gpurun_out/c_test.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/c_bench_base.log
(DPC_STEM_FUSED=1 timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/c_bench_stemfused.log
cd /tmp && export TMPDIR=/tmp
(DPC_STEM_FUSED=1 timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/c_rocprof.log
cd $R
f=$(ls gpurun_out/prof_c/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/c_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_c/*.db
tail -6 gpurun_out/c_test.log
for f in base stemfused; do python -c "
import json; d=json.loads(open('gpurun_out/c_bench_$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['hbm_family']['ms_per_step'], d['hbm_family']['frac'], d['wgrad_kernel'])"; done
grep -E "stem|pool|igemm_kernel|halo" gpurun_out/c_kernel_stats.txt | cut -c1-150
