#!/bin/bash
# HBM traffic counters of one bench step (two separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one
# pass); summarised by scripts/pmc_traffic.py into profiles/<tag>_pmc_traffic.json
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-graph"
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o pmc -- $B 2>&1 | tail -3) > $R/gpurun_out/pmc_$c.log
done
cd $R; python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/*.db gpurun_out/pmc_WRITE_SIZE/*.db ${PMC_TAG:-r01}
