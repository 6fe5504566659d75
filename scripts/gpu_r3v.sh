#!/bin/bash
# round 3, session v: weight gradients on a side stream beside the next unit's BatchNorm backward
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py tests/test_block_grads_gpu.py tests/test_configs_gpu.py tests/test_checkpoint_gpu.py tests/test_lc_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/v_test.log
B="python bench.py --no-cpu-baseline --no-also --steps 100"
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/v_bench_cfg2.log
(DPC_WGRAD_STREAM=0 timeout 300 $B 2>&1 | tail -1) > gpurun_out/v_bench_cfg2_onestream.log
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/v_bench_cfg2_b.log
B4="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 30 --config"
for c in cfg4 cfg5; do
  (timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/v_bench_$c.log
  (DPC_WGRAD_STREAM=0 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/v_bench_${c}_onestream.log
done
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 3 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1) > gpurun_out/v_bench_torchrun.log
cat gpurun_out/v_test.log
for f in gpurun_out/v_bench_*.log; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('final_loss'))
except Exception as e: print('ERR', open(sys.argv[1]).read()[-400:])
P
)"; done
