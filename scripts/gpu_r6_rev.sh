#!/bin/bash
# round 6: cache-aware traversal order of the forward element-wise kernels (DPC_BN_APPLY_REV / DPC_POOL_FWD_REV / DPC_PACK_REV), A/B in
# alternating fresh processes on one box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r06_rev_ab.txt; : > $O
B="python bench.py --no-cpu-baseline --no-also --no-roofline --steps 100"
run() { echo "== $1" >> $O; (env $1 timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])") >> $O 2>&1; }
for rep in 1 2; do
run "DPC_X=0"
run "DPC_BN_APPLY_REV=1"
run "DPC_POOL_FWD_REV=1"
run "DPC_PACK_REV=1"
run "DPC_BN_APPLY_REV=1 DPC_POOL_FWD_REV=1 DPC_PACK_REV=1"
run "DPC_BN_NT_MB=100000"
run "DPC_BN_NT_MB=100000 DPC_BN_APPLY_REV=1 DPC_POOL_FWD_REV=1 DPC_PACK_REV=1"
done
cat $O
