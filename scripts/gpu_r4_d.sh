#!/bin/bash
# round 4, session D: forward two-stream probe, score GEMM after the grid fix, kernel tier with the tightened tolerance / mutation
# checks / aliasing cases, the one test that failed in session C (repeated), one bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 300 python scripts/probes/fwd_overlap_probe.py 20 2>&1 | grep -v amdgpu.ids) > gpurun_out/d_fwd_overlap.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "gru_chain\|score_fwd\|score_bwd") > gpurun_out/d_head.log
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/d_test_kernels.log
(for i in 1 2 3; do timeout 300 python -m pytest tests/test_graph_rccl_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4; done) > gpurun_out/d_test_graph.log
(timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_block_grads_gpu.py tests/test_two_stream_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/d_test_engine.log
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/d_bench_cfg2.log
cat gpurun_out/d_fwd_overlap.log gpurun_out/d_head.log gpurun_out/d_test_kernels.log gpurun_out/d_test_graph.log gpurun_out/d_test_engine.log
python -c "import json; d=json.loads(open('gpurun_out/d_bench_cfg2.log').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('score_gemm',{}).get('frac'), d.get('score_gemm',{}).get('us_per_step'))"
