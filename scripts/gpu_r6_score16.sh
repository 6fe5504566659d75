#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "bf16_logits or gemm_nt or ce_topk" 2>&1 | tail -5) > gpurun_out/r06_s16_tests.log
(timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider -k "bf16_logits or full_batch or cfg5" 2>&1 | tail -15) >> gpurun_out/r06_s16_tests.log
cat gpurun_out/r06_s16_tests.log
STEPS=100 bash scripts/gpu_ab.sh r06_score16_ab.txt DPC_SCORE_BF16=0 DPC_X=1
(timeout 300 python bench.py --no-cpu-baseline --no-also --pmc off 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['score_gemm'])") > gpurun_out/r06_s16_line.txt 2>&1
cat gpurun_out/r06_s16_line.txt
