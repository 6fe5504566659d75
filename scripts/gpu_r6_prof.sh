#!/bin/bash
# round 6: rocprofv3 kernel trace of the cfg2 step (kernel by kernel, 2 streams): stats + timeline
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out
c=${1:-cfg2}; n=4
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --no-graph --steps $((n-1)) --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -2) > $R/gpurun_out/p_rocprof_$c.log
f=$(ls $R/gpurun_out/prof_$c/*.db $R/gpurun_out/prof_$c/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then
  (cd $R && python scripts/rocpd_stats.py $f $n > gpurun_out/r06_${c}_kernel_stats.txt 2>&1)
  (cd $R && python scripts/timeline.py $f "." > gpurun_out/r06_${c}_step_timeline.txt 2>&1)
fi
rm -rf $R/gpurun_out/prof_$c
cd $R; head -45 gpurun_out/r06_${c}_kernel_stats.txt
