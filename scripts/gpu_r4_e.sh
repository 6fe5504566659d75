#!/bin/bash
# round 4, session E: bf16x6 (f32 tensors, contractions on the bf16 pipe) against the reference goldens + its step rate; the
# finalize-before-fork ordering A/B; kernel tier after the tolerance / mutation / score-kernel-selection changes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -rP -p no:cacheprovider -k "eval_score_vs_reference or train_step_vs_reference" 2>&1 | grep -E "passed|failed|max \|score|Error|assert" | tail -20) > gpurun_out/e_test_x6.log
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6) > gpurun_out/e_test_kernels.log
(timeout 300 python bench.py --dtype bf16x6 --steps 6 --warmup 2 --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/e_bench_x6.log
(timeout 300 python bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/e_bench_f32.log
(DPC_EARLY_FINALIZE=0 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1) > gpurun_out/e_bench_late.log
(DPC_EARLY_FINALIZE=1 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1) > gpurun_out/e_bench_early.log
(DPC_EARLY_FINALIZE=0 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1) > gpurun_out/e_bench_late2.log
(DPC_EARLY_FINALIZE=1 timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1) > gpurun_out/e_bench_early2.log
(timeout 600 python -m pytest tests/test_two_stream_gpu.py tests/test_cotenant_gpu.py tests/test_graph_rccl_gpu.py tests/test_checkpoint_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/e_test_sched.log
cat gpurun_out/e_test_x6.log gpurun_out/e_test_kernels.log gpurun_out/e_test_sched.log
python - <<'P'
import json
for f in ("e_bench_x6","e_bench_f32","e_bench_late","e_bench_early","e_bench_late2","e_bench_early2"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.log").read().strip().splitlines()[-1]); print(f, d['dtype'], d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('wgrad_kernel'))
    except Exception as e: print(f, "ERR", open(f"gpurun_out/{f}.log").read()[-400:])
P
