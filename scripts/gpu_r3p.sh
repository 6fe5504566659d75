#!/bin/bash
# round 3, session p: 64-column strided input-gradient on the loader/compute kernel; step-level effect
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_block_grads_gpu.py -x -q -k "dgrad or block" 2>&1 | tail -3) > gpurun_out/p_test_dgrad.log
S="l2s l3s l4s l2s224 l3s224"
(echo "# default"; timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 (round-3 start)"; DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# default again"; timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F) > gpurun_out/p_conv_bench.txt
B="python bench.py --no-cpu-baseline --no-also --steps 100"
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/p_bench_cfg2.log
(DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 300 $B 2>&1 | tail -1) > gpurun_out/p_bench_cfg2_old_dgrad.log
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/p_bench_cfg2_b.log
B4="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 30 --config"
for c in cfg4 cfg5; do
  (timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/p_bench_$c.log
  (DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/p_bench_${c}_old_dgrad.log
done
cat gpurun_out/p_test_dgrad.log gpurun_out/p_conv_bench.txt
for f in gpurun_out/p_bench_*.log; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
except Exception as e: print('ERR', open(sys.argv[1]).read()[-300:])
P
)"; done
