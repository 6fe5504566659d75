#!/bin/bash
# round 3, session t: full GPU tier + bench on the current build
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > gpurun_out/t_test.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/t_smoke.log
(timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/t_bench.log
cat gpurun_out/t_test.log gpurun_out/t_smoke.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/t_bench.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['score_gemm']['frac'], d['hbm_family']['ms_per_step'], {k:(v['value'],v['ms_per_step']) for k,v in d.get('also',{}).items()})
P
