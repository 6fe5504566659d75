#!/bin/bash
# round 5 -> profiles/r05_clock_ramp.txt: does a fresh process measure its steady step time?  (clock ramp probe + the driver's flags against a long run)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
O=gpurun_out/r05_clock_ramp.txt
(timeout 300 python scripts/probes/clock_ramp_probe.py 300 2>&1 | grep -v amdgpu.ids) > $O
for k in 20 20 200; do
  line=$(timeout 300 python bench.py --steps $k --warmup 5 --no-also --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "bench.py --steps $k --warmup 5: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'])" "$line")" >> $O
done
cat $O
