#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 200 python scripts/stem_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/e_stem_bench.txt
(timeout 900 python -m pytest tests/test_block_grads_gpu.py tests/test_kernels_gpu.py -m gpu -q -rP -p no:cacheprovider -k "block or reduction or splitk or stem_wgrad" 2>&1 | grep -E "passed|failed|Error|assert|worst|fold_c1" | tail -30) > gpurun_out/e_test.log
(DPC_STEM_FUSED=1 timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/e_bench_stemfused.log
cat gpurun_out/e_stem_bench.txt; cat gpurun_out/e_test.log
python -c "
import json; d=json.loads(open('gpurun_out/e_bench_stemfused.log').read().strip().splitlines()[-1]); print('stemfused', d['value'], d['ms_per_step'], d['roofline']['frac'], d['hbm_family']['ms_per_step'], d['wgrad_kernel'])"
