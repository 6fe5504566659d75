#!/bin/bash
# round-2 GPU session A: full gpu test tier, smoke, bench lines for cfg2/cfg4/cfg5 + f32 parity mode, kernel-trace profiles
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1700 python -m pytest tests -m gpu -q -rP -p no:cacheprovider 2>&1) > gpurun_out/test_gpu_full.log
grep -E "passed|failed|error|bf16 anchor|resnet|norm err" gpurun_out/test_gpu_full.log | tail -120 > gpurun_out/test_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log
(timeout 600 python bench.py 2>&1 | tail -3) > gpurun_out/bench_cfg2.log
(timeout 300 python bench.py --no-graph --no-cpu-baseline --steps 100 2>&1 | tail -3) > gpurun_out/bench_cfg2_nograph.log
(timeout 400 python bench.py --config cfg4 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_cfg4.log
(timeout 400 python bench.py --config cfg5 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_cfg5.log
(timeout 400 python bench.py --dtype f32 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -3) > gpurun_out/bench_cfg2_f32.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > gpurun_out/bench_torchrun.log
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02a -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -5) > $R/gpurun_out/rocprof_a.log
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02a_cfg4 -o bench -- python $R/bench.py --config cfg4 --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -5) > $R/gpurun_out/rocprof_a4.log
cd $R
for d in prof_r02a prof_r02a_cfg4; do
  f=$(ls gpurun_out/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/${d}_stats.txt 2>&1
done
f=$(ls gpurun_out/prof_r02a/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/timeline.py $f "." > gpurun_out/prof_r02a_timeline.txt 2>&1
rm -rf gpurun_out/prof_r02a/*.db gpurun_out/prof_r02a_cfg4/*.db 2>/dev/null  # keep the merge-back small
ls gpurun_out | head -50
