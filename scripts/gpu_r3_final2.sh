#!/bin/bash
# round-3 closing session after the LDS claim of the persistent kernels: GPU tier, default bench line, kernel trace, PMC traffic
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4) > gpurun_out/zz_test.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/zz_smoke.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/zz_bench_cfg2.log
(DPC_WGRAD_STREAM=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/zz_bench_cfg2_one_stream.log
(timeout 200 python scripts/probes/corun_probe.py l3 10000 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/zz_corun_l3.txt
(timeout 200 python scripts/probes/corun_probe.py l1 10000 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/zz_corun_l1.txt
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03 -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/zz_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r03/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r03_stats.txt 2>&1 && python scripts/timeline.py $f "." > gpurun_out/prof_r03_timeline.txt 2>&1
rm -rf gpurun_out/prof_r03/*.db
PMC_TAG=r03 bash scripts/gpu_pmc_traffic.sh > gpurun_out/zz_pmc_traffic.log 2>&1
cp profiles/r03_pmc_traffic.json profiles/r03_pmc_traffic_per_kernel.txt gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE/*.db gpurun_out/pmc_WRITE_SIZE/*.db
cat gpurun_out/zz_test.log gpurun_out/zz_smoke.log gpurun_out/zz_corun_l3.txt gpurun_out/zz_corun_l1.txt; for f in gpurun_out/zz_bench_*.log; do echo "$f: $(tail -1 $f | cut -c1-170)"; done
