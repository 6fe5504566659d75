#!/bin/bash
# round 3, session q: generic implicit GEMM with two chunks in flight (A/B against a -DDPC_IGEMM_PIPE2=0 build of the same sources)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
P1=$R/scripts/probes/libdpc_pipe1.so
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/q_test_kernels.log
S="l2s l3s l4s l2s224 l3s224"
(echo "# two chunks in flight"; timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# one chunk in flight (-DDPC_IGEMM_PIPE2=0)"; DPC_BENCH_LIB=$P1 timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F
 echo "# two chunks in flight, DPC_PARITY_ILV=0"; DPC_PARITY_ILV=0 timeout 120 python scripts/conv_bench.py l2s l2s224 --ops dgrad 2>&1 | $F
 echo "# two chunks in flight, again"; timeout 120 python scripts/conv_bench.py $S --ops dgrad 2>&1 | $F) > gpurun_out/q_conv_bench.txt
(echo "# two chunks in flight"; REPS=20 timeout 200 python scripts/head_bench.py 2>&1 | $F | grep -i "score GEMM\|d_pred\|d_finf\|CE"
 echo "# one chunk in flight"; DPC_BENCH_LIB=$P1 REPS=20 timeout 200 python scripts/head_bench.py 2>&1 | $F | grep -i "d_pred split") > gpurun_out/q_head_bench.txt
B="python bench.py --no-cpu-baseline --no-also --steps 100"
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/q_bench_cfg2.log
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/q_bench_cfg2_b.log
(timeout 300 $B --dtype f32 --steps 20 2>&1 | tail -1) > gpurun_out/q_bench_cfg2_f32.log
B4="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 30 --config"
for c in cfg4 cfg5; do (timeout 300 $B4 $c 2>&1 | tail -1) > gpurun_out/q_bench_$c.log; done
cat gpurun_out/q_test_kernels.log gpurun_out/q_conv_bench.txt gpurun_out/q_head_bench.txt
for f in gpurun_out/q_bench_*.log; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('score_gemm',{}).get('frac'))
except Exception as e: print('ERR', open(sys.argv[1]).read()[-300:])
P
)"; done
