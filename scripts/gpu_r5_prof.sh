#!/bin/bash
# round 5: rocprofv3 kernel trace (stats; timeline for cfg2) of cfg2 / cfg4 / cfg5 on the current build
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for c in cfg2 cfg4 cfg5; do
  n=3; [ $c = cfg2 ] && n=4
  (timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --no-graph --steps $((n-1)) --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -2) > $R/gpurun_out/p_rocprof_$c.log
  f=$(ls $R/gpurun_out/prof_$c/*.db $R/gpurun_out/prof_$c/*/*.db 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    (cd $R && python scripts/rocpd_stats.py $f $n > gpurun_out/r05_${c}_kernel_stats.txt 2>&1)
    [ $c = cfg2 ] && (cd $R && python scripts/timeline.py $f "." > gpurun_out/r05_cfg2_step_timeline.txt 2>&1)
  fi
  rm -rf $R/gpurun_out/prof_$c
done
cd $R; head -30 gpurun_out/r05_cfg4_kernel_stats.txt
