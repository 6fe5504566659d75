#!/bin/bash
# round-3 session A: full gpu tier, smoke, default bench line (with the cfg4 / cfg5 side measurements), MFMA-rate probe,
# kernel stats of cfg2
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30) > gpurun_out/a_test.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/a_smoke.log
(timeout 900 python bench.py 2>&1 | tail -2) > gpurun_out/a_bench.log
(timeout 200 python scripts/probes/mfma_rate.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/a_mfma_rate.txt
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $R/gpurun_out/prof_mfma -o p -- python $R/scripts/probes/mfma_rate.py 2>&1 | tail -3) > $R/gpurun_out/a_mfma_pmc.log
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_a -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/a_rocprof.log
cd $R
f=$(ls gpurun_out/prof_a/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/a_kernel_stats.txt 2>&1
f=$(ls gpurun_out/prof_mfma/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/pmc_table.py $f > gpurun_out/a_mfma_pmc.txt 2>&1
rm -rf gpurun_out/prof_a/*.db gpurun_out/prof_mfma/*.db
tail -3 gpurun_out/a_test.log; cat gpurun_out/a_smoke.log; cat gpurun_out/a_bench.log | cut -c1-1500; cat gpurun_out/a_mfma_rate.txt
