#!/bin/bash
# round-2 GPU session B: new kernels (fused ConvGRU recurrence, dropout, adam_dev) + RCCL/graph diagnosis + bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "gru_chain or dropout or adam" 2>&1 | tail -30) > gpurun_out/b_kernels.log
(timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py tests/test_configs_gpu.py -q -rP -p no:cacheprovider 2>&1) > gpurun_out/b_engine_full.log
grep -E "passed|failed|error|bf16 anchor|resnet|Error|assert" gpurun_out/b_engine_full.log | tail -60 > gpurun_out/b_engine.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 scripts/diag_rccl_graph.py 2>&1 | grep -v "^W\|amdgpu.ids" | tail -20) > gpurun_out/b_diag_rccl.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/b_smoke.log
(timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/b_bench_cfg2.log
(timeout 300 python bench.py --config cfg5 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/b_bench_cfg5.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02b -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/b_rocprof.log
cd $R
f=$(ls gpurun_out/prof_r02b/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/prof_r02b_stats.txt 2>&1 && python scripts/timeline.py $f "gru|score|igemm_kernel|ce_|colsum|wgrad2|reduce_unpack|copy2d|transpose|tpool|dropout|step_adv|adam" > gpurun_out/prof_r02b_timeline.txt 2>&1
rm -rf gpurun_out/prof_r02b/*.db
ls gpurun_out | head -40
