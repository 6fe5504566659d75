#!/bin/bash
# A/B of environment switches on the cfg2 step, alternating fresh processes on one box: scripts/gpu_ab.sh OUT "ENV1" "ENV2" ...  (each twice)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/$1; shift; : > $O
B="python bench.py --no-cpu-baseline --no-also --no-roofline --steps ${STEPS:-100} ${BARGS:-}"
for rep in 1 2; do
for e in "$@"; do
echo "== $e" >> $O; (env $e timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])") >> $O 2>&1
done; done
cat $O
