#!/bin/bash
# round 5 -> profiles/r05_score_gemm_wgs.txt: score GEMM workgroup count (one wave of workgroups against rounds 3-4's rounding up), isolated and in the step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
O=gpurun_out/r05_score_gemm_wgs.txt; : > $O
for w in 559 512 559 512; do
  echo "== DPC_SCORE_GEMM_WGS=$w (559: 11 column splits = 528 workgroups at R = 6144, the old plan; 512: 10 splits = 480)" >> $O
  (DPC_SCORE_GEMM_WGS=$w REPS=20 timeout 300 python scripts/head_bench.py 2>&1 | grep -E "score GEMM") >> $O
done
echo "== DPC_SCORE_GEMM2=2 (8-wave form wherever the shape allows)" >> $O
(DPC_SCORE_GEMM2=2 REPS=20 timeout 300 python scripts/head_bench.py 2>&1 | grep -E "score GEMM") >> $O
for w in 559 512 559 512; do
  line=$(DPC_SCORE_GEMM_WGS=$w timeout 300 python bench.py --steps 40 --warmup 5 --no-also --no-cpu-baseline --pmc off 2>/dev/null | tail -1)
  echo "in the step, DPC_SCORE_GEMM_WGS=$w: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'], 'score', d['score_gemm']['us_per_step'], d['score_gemm']['frac'])" "$line")" >> $O
done
cat $O
