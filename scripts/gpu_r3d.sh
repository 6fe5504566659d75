#!/bin/bash
# round-3 session D: stem micro-benchmark, head micro-benchmark, tests of the head kernels, bench A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 200 python scripts/stem_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/d_stem_bench.txt
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/d_head_bench.txt
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -k "stem or gemm or ce_topk or score or full_batch or train_step or golden" 2>&1 | tail -15) > gpurun_out/d_test.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/d_bench_base.log
(DPC_STEM_FUSED=1 timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/d_bench_stemfused.log
cat gpurun_out/d_stem_bench.txt; cat gpurun_out/d_head_bench.txt; tail -6 gpurun_out/d_test.log
for f in base stemfused; do python -c "
import json; d=json.loads(open('gpurun_out/d_bench_$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'], d['hbm_family']['ms_per_step'], d['score_gemm']['frac'], d['score_gemm']['us_per_step'])"; done
