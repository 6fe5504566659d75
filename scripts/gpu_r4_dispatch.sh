#!/bin/bash
# round 4: kernels whose run time was workgroup dispatch (~5 ns per workgroup): weight repack (19 392 workgroups), few-slab reductions (12 288)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "bn or reduce_unpack or gemm" 2>&1 | tail -3) > gpurun_out/l_test.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids | grep "R=6144\|R=15680") > gpurun_out/l_head.log
for k in "512 1024" "4096 256" "512 1024" "4096 256"; do set -- $k; (DPC_PACK_BLOCK_ELEMS=$1 DPC_PACK_BLOCK_CAP=$2 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pack elems/wg=$1 cap=$2', d['value'], d['ms_per_step'], 'score', d['score_gemm']['frac'], d['score_gemm']['us_per_step'])"); done > gpurun_out/l_bench.log
cat gpurun_out/l_test.log gpurun_out/l_head.log gpurun_out/l_bench.log
