#!/bin/bash
# round 4 closing: rocprofv3 kernel trace of the cfg2 step on the final build (per-kernel statistics + one step's timeline)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_z -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/z_rocprof.log
cd $R
f=$(ls gpurun_out/prof_z/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f 4 > gpurun_out/z_kernel_stats_cfg2.txt 2>&1 && python scripts/timeline.py $f "." > gpurun_out/z_timeline_cfg2.txt 2>&1
rm -rf gpurun_out/prof_z
head -30 gpurun_out/z_kernel_stats_cfg2.txt; tail -2 gpurun_out/z_timeline_cfg2.txt; tail -1 gpurun_out/z_rocprof.log | cut -c1-200
