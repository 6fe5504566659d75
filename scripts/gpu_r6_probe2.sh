#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python scripts/probes/halo_epi_probe.py 2>&1 | grep -E "full kernel|no MFMA loop  |no DMA, no epilogue") > gpurun_out/r06_halo_epi_probe_after.txt
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "halo or dgrad or epi or conv" 2>&1 | tail -3) > gpurun_out/r06_p2_tests.log
(timeout 900 python -m pytest tests/test_block_grads_gpu.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5) >> gpurun_out/r06_p2_tests.log
for i in 1 2; do (timeout 200 python bench.py --no-cpu-baseline --no-also --no-roofline --steps 100 2>&1 | tail -1 | cut -c1-200) >> gpurun_out/r06_p2_bench.txt; done
cat gpurun_out/r06_halo_epi_probe_after.txt gpurun_out/r06_p2_tests.log gpurun_out/r06_p2_bench.txt
