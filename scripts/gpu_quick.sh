#!/bin/bash
# quick A/B on the GPU box: kernel parity tests for the touched kernels + two short bench runs
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "${QUICK_K:-wgrad or stem}" > gpurun_out/quick_test.log 2>&1
tail -3 gpurun_out/quick_test.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/quick_bench_new.log 2>&1
tail -1 gpurun_out/quick_bench_new.log | cut -c1-200; tail -1 gpurun_out/quick_bench_new.log | grep -o '"wgrad_kernel.*'
if [ -n "$QUICK_AB" ]; then
  env $QUICK_AB python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/quick_bench_old.log 2>&1
  tail -1 gpurun_out/quick_bench_old.log | cut -c1-200; tail -1 gpurun_out/quick_bench_old.log | grep -o '"wgrad_kernel.*'
fi
