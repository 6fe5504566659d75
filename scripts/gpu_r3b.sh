#!/bin/bash
# round-3 session B: full gpu tier, A/B of the fused backward epilogues (DPC_FOLD), kernel stats
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/b_test.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/b_bench_fold1.log
(DPC_FOLD=0 timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/b_bench_fold0.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/b_rocprof.log
cd $R
f=$(ls gpurun_out/prof_b/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/b_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_b/*.db
tail -8 gpurun_out/b_test.log
for f in fold1 fold0; do python -c "
import json; d=json.loads(open('gpurun_out/b_bench_$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['hbm_family']['ms_per_step'], d['hbm_family']['frac'])"; done
head -40 gpurun_out/b_kernel_stats.txt | cut -c1-150
