#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 200 python scripts/stem_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/f_stem_bench.txt
(timeout 900 python -m pytest tests/test_block_grads_gpu.py -m gpu -q -rP -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|worst|fold_c1" | tail -30) > gpurun_out/f_test.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/f_bench.log
cat gpurun_out/f_stem_bench.txt; cat gpurun_out/f_test.log
python -c "
import json; d=json.loads(open('gpurun_out/f_bench.log').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['hbm_family']['ms_per_step'], d['wgrad_kernel'], d['score_gemm']['frac'])"
