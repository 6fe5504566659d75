#!/bin/bash
# round 5: BatchNorm-apply inside the consuming layer1 conv (DPC_BNIN): kernel cases on the device, then the alternating A/B in the step
# -> profiles/r05_bnin_ab.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "bnin or conv_fwd" 2>&1 | tail -4) > gpurun_out/bnin_tests.log
cat gpurun_out/bnin_tests.log
O=gpurun_out/r05_bnin_ab.txt; : > $O
arm() {
  line=$(DPC_BNIN=$2 timeout 300 python bench.py --config $1 --steps $3 --warmup 5 --no-also --no-cpu-baseline --pmc off 2>/dev/null | tail -1)
  echo "$1 DPC_BNIN=$2 $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'], 'conv', d['roofline']['ms_per_step'], d['roofline']['frac'], 'hbm_family', d['hbm_family']['ms_per_step'], d['hbm_family']['launches_per_step'])" "$line")" | tee -a $O
}
for i in 1 2 3; do arm cfg2 1 40; arm cfg2 0 40; done
for i in 1 2; do arm cfg4 1 12; arm cfg4 0 12; done
