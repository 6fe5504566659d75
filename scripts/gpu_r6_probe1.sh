#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python scripts/probes/halo_epi_probe.py 2>&1 | tail -40) > gpurun_out/r06_halo_epi_probe.txt
(DPC_FOLD=0 timeout 200 python bench.py --no-cpu-baseline --no-also --no-roofline --steps 100 2>&1 | tail -1 | cut -c1-200) > gpurun_out/r06_fold0.txt
(timeout 200 python bench.py --no-cpu-baseline --no-also --no-roofline --steps 100 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r06_fold0.txt
cat gpurun_out/r06_halo_epi_probe.txt gpurun_out/r06_fold0.txt
