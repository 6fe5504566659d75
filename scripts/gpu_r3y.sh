#!/bin/bash
# round 3, session y: two streams with the main stream waiting for the side stream before every input-gradient (DPC_SIDE_QUIET=1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 100"
(DPC_WGRAD_STREAM=1 DPC_SIDE_QUIET=1 timeout 300 $B 2>&1 | tail -1) > gpurun_out/y_bench_cfg2_quiet.log
(DPC_WGRAD_STREAM=0 timeout 300 $B 2>&1 | tail -1) > gpurun_out/y_bench_cfg2_one.log
(DPC_WGRAD_STREAM=1 timeout 300 $B 2>&1 | tail -1) > gpurun_out/y_bench_cfg2_two.log
(DPC_WGRAD_STREAM=1 DPC_SIDE_QUIET=1 timeout 300 $B 2>&1 | tail -1) > gpurun_out/y_bench_cfg2_quiet_b.log
B4="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 30 --config"
for c in cfg4 cfg5; do
  (DPC_WGRAD_STREAM=1 DPC_SIDE_QUIET=1 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/y_bench_${c}_quiet.log
  (DPC_WGRAD_STREAM=0 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/y_bench_${c}_one.log
done
for f in gpurun_out/y_bench_*.log; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('final_loss'))
except Exception as e: print('ERR', open(sys.argv[1]).read()[-400:])
P
)"; done
