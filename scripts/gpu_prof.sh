#!/bin/bash
# kernel-trace profile of a short bench run; summary printed by scripts/rocpd_stats.py
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/prof_q
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/prof_q.log
cd $R; python scripts/rocpd_stats.py gpurun_out/prof_q/*.db | head -${PROF_LINES:-24}
