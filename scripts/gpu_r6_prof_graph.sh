#!/bin/bash
# round 6: rocprofv3 kernel trace of the cfg2 step AS REPLAYED (hipGraph, two streams): the timeline the headline number runs
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out
c=${1:-cfg2}
(timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/profg_$c -o bench -- python $R/bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -2) > $R/gpurun_out/pg_rocprof_$c.log
f=$(ls $R/gpurun_out/profg_$c/*.db $R/gpurun_out/profg_$c/*/*.db 2>/dev/null | head -1)
if [ -n "$f" ]; then
  (cd $R && python scripts/timeline.py $f "." > gpurun_out/r06_${c}_graph_timeline.txt 2>&1)
fi
rm -rf $R/gpurun_out/profg_$c
cd $R; tail -5 gpurun_out/pg_rocprof_$c.log; tail -3 gpurun_out/r06_${c}_graph_timeline.txt
