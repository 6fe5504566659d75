#!/bin/bash
# refresh the PMC traffic summary (family pattern now includes the plane kernel) + one bench line that reads it
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
PMC_TAG=r02 bash scripts/gpu_pmc_traffic.sh > gpurun_out/l_pmc_traffic.log 2>&1
cp profiles/r02_pmc_traffic.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE/*.db gpurun_out/pmc_WRITE_SIZE/*.db
(timeout 600 python bench.py 2>&1 | tail -1) > gpurun_out/l_bench_cfg2.log
cat gpurun_out/r02_pmc_traffic.json; tail -c 1500 gpurun_out/l_bench_cfg2.log
