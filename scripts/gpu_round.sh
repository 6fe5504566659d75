cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/env.log
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150) > gpurun_out/test_gpu.log
echo "pytest done" >> gpurun_out/test_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -20) > gpurun_out/smoke.log
(timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -20) > gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -15) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r01 | head -30 >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -5) > gpurun_out/bench_torchrun.log
