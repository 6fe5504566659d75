#!/bin/bash
# HISTORICAL (round 4): sets DPC_SCORE_GEMM_COUNTED / DPC_EARLY_FINALIZE, switches that were removed in round 5 -- those A/B arms now run
# identical code.  Kept for the record of what round 4 measured; the current recipes are scripts/gpu_r6_*.sh.
# round 4: the probes behind DESIGN section 9.2 / 9.4 / section 5 (sessions D and E of the round): head kernel timings, the forward
# two-stream probe, the finalize-before-fork A/B, bf16x6 against the reference goldens and its step rate beside exact f32
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/pr_head.log
(timeout 300 python scripts/probes/fwd_overlap_probe.py 20 2>&1 | grep -v amdgpu.ids) > gpurun_out/pr_fwd_overlap.log
for v in 0 1 0 1; do (DPC_EARLY_FINALIZE=$v timeout 300 python bench.py --steps 60 --no-cpu-baseline --no-also --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DPC_EARLY_FINALIZE=$v', d['value'], d['ms_per_step'])"); done > gpurun_out/pr_early_finalize.log
(timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -rP -p no:cacheprovider -k "eval_score_vs_reference or train_step_vs_reference" 2>&1 | grep -E "passed|failed|max \|score" | tail -12) > gpurun_out/pr_x6_goldens.log
for dt in f32 bf16x6; do (timeout 300 python bench.py --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['wgrad_kernel'])"); done > gpurun_out/pr_x6_rate.log
cat gpurun_out/pr_*.log
