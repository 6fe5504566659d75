#!/bin/bash
# (CFG=cfg4 / cfg5 picks another BASELINE configuration)
# A/B of alternative builds of the kernel library (dpc_amd/alt/libdpc_hip_<tag>.so copied over the product library inside the gpurun
# sandbox), cfg2 step, alternating fresh processes: scripts/gpu_altlib_ab.sh OUT tag1 tag2 ...   ("base" = the product library)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/$1; shift; : > $O
cp dpc_amd/libdpc_hip.so /tmp/libdpc_base.so
B="python bench.py --config ${CFG:-cfg2} --no-cpu-baseline --no-also --no-roofline --steps ${STEPS:-100}"
for rep in 1 2; do
for t in "$@"; do
  if [ "$t" = base ]; then cp /tmp/libdpc_base.so dpc_amd/libdpc_hip.so; else cp dpc_amd/alt/libdpc_hip_$t.so dpc_amd/libdpc_hip.so; fi
  echo "== $t" >> $O; (timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])") >> $O 2>&1
done; done
cp /tmp/libdpc_base.so dpc_amd/libdpc_hip.so
cat $O
