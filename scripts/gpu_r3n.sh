#!/bin/bash
# round 3, session n: strided input-gradients on the loader/compute kernel + interleaved classes; score GEMM store forms; 224^2 A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
(timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dgrad" 2>&1 | tail -5) > gpurun_out/n_test_dgrad.log
(timeout 600 python -m pytest tests/test_block_grads_gpu.py tests/test_plan.py -x -q 2>&1 | tail -5) > gpurun_out/n_test_block.log
S="l2s l3s l4s l2s224 l3s224"
(echo "# default"; timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_PARITY_ILV=0"; DPC_PARITY_ILV=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_IGEMM_WS_PAR=0"; DPC_IGEMM_WS_PAR=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 (round-3 start)"; DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F) > gpurun_out/n_conv_bench.txt
(echo "# default (row-major dword stores)"; REPS=20 timeout 200 python scripts/head_bench.py 2>&1 | $F | grep -i "score GEMM\|d_pred\|d_finf\|CE"
 echo "# DPC_SCORE_GEMM_SWAP=1"; DPC_SCORE_GEMM_SWAP=1 REPS=20 timeout 200 python scripts/head_bench.py 2>&1 | $F | grep -i "score GEMM") > gpurun_out/n_head_bench.txt
B="python bench.py --no-cpu-baseline --no-also --steps 100"
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/n_bench_cfg2.log
(DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 300 $B 2>&1 | tail -1) > gpurun_out/n_bench_cfg2_old_dgrad.log
B4="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 30 --config"
for c in cfg4 cfg5; do
  (timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/n_bench_$c.log
  (DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/n_bench_${c}_old_dgrad.log
  (DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 DPC_FOLD=0 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/n_bench_${c}_nofold.log
  (DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 DPC_STEM_FUSED=0 timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/n_bench_${c}_nostemfused.log
done
cat gpurun_out/n_test_dgrad.log gpurun_out/n_test_block.log gpurun_out/n_conv_bench.txt gpurun_out/n_head_bench.txt
for f in gpurun_out/n_bench_*.log; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
except Exception as e: print('ERR', open(sys.argv[1]).read()[-300:])
P
)"; done
