#!/bin/bash
# round 5: alternating A/B of the one-stream / two-stream schedules with HIP_FORCE_DEV_KERNARG=1 (the package default), fresh process per arm
#  -> profiles/r05_two_stream_ab.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
AB=gpurun_out/r05_two_stream_ab.txt
echo "# alternating A/B, fresh process per arm, HIP_FORCE_DEV_KERNARG default (=1); bench.py --no-also --no-roofline --no-cpu-baseline" > $AB
arm() {  # config steps stream
  line=$(DPC_WGRAD_STREAM=$3 timeout 300 python bench.py --config $1 --steps $2 --warmup 5 --no-also --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$1 streams=$3 $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'], d['config']['streams'])" "$line")" | tee -a $AB
}
for i in 1 2 3 4 5; do arm cfg2 40 1; arm cfg2 40 0; done
for i in 1 2 3; do arm cfg4 12 1; arm cfg4 12 0; done
for i in 1 2 3; do arm cfg5 12 1; arm cfg5 12 0; done
python - <<'P' | tee -a gpurun_out/r05_two_stream_ab.txt
import collections, statistics
rows = collections.defaultdict(list)
for l in open('gpurun_out/r05_two_stream_ab.txt'):
    p = l.split()
    if len(p) >= 3 and p[0].startswith('cfg'):
        rows[(p[0], p[1])].append(float(p[2]))
for cfg in ('cfg2', 'cfg4', 'cfg5'):
    a, b = rows[(cfg, 'streams=1')], rows[(cfg, 'streams=0')]
    if a and b:
        print(f"# {cfg}: two streams median {statistics.median(a):.3f} ms (n={len(a)}), one stream median {statistics.median(b):.3f} ms (n={len(b)}), "
              f"one/two = {statistics.median(b)/statistics.median(a):.4f}")
P
