#!/bin/bash
# round 6 closing session: GPU tier, smoke, the default bench line (live PMC, also.*, cpu_baseline), bench under torch.distributed.run
# (1 rank), rocprofv3 kernel stats of cfg2 / cfg4 / cfg5 (kernel by kernel) + the timeline of the REPLAYED cfg2 graph, per-kernel
# HBM / MFMA counters of the one-stream step, PMC traffic per kernel, two-stream == one-stream bit identity
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/z_test_full.log 2>&1; echo "rc=$?" >> gpurun_out/z_test_full.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > gpurun_out/z_smoke.log
(timeout 1200 python bench.py 2>gpurun_out/z_bench.err | tail -1) > gpurun_out/z_bench_cfg2.log
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 40 --warmup 3 --no-cpu-baseline --no-roofline --schedules 2>&1 | tail -1) > gpurun_out/z_bench_torchrun.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/z_head.log
(timeout 600 python scripts/stream_stress.py cfg2 300 1 2>&1 | tail -4; timeout 400 python scripts/stream_stress.py cfg5 60 1 2>&1 | tail -3) > gpurun_out/z_two_stream_identity.log
# kernel stats (kernel by kernel, two streams) of the three configurations
cd /tmp && export TMPDIR=/tmp
for c in cfg2 cfg4 cfg5; do
  n=3; [ $c = cfg2 ] && n=4
  (timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o bench -- python $R/bench.py --config $c --no-graph --steps $((n-1)) --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -2) > $R/gpurun_out/p_rocprof_$c.log
  f=$(ls $R/gpurun_out/prof_$c/*.db $R/gpurun_out/prof_$c/*/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $R && python scripts/rocpd_stats.py $f $n > gpurun_out/r06_${c}_kernel_stats.txt 2>&1)
  rm -rf $R/gpurun_out/prof_$c
done
cd $R
bash scripts/gpu_r6_prof_graph.sh cfg2 > /dev/null 2>&1
# per-kernel counters (one stream)
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-graph --pmc off"
export DPC_WGRAD_STREAM=0
(timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/rf_trace -o t -- $B 2>&1 | tail -1) > $R/gpurun_out/rf.log
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/rf_$c -o t -- $B 2>&1 | tail -1) >> $R/gpurun_out/rf.log
done
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/rf_sq -o t -- $B 2>&1 | tail -1) >> $R/gpurun_out/rf.log
unset DPC_WGRAD_STREAM
cd $R
db() { ls gpurun_out/$1/*.db gpurun_out/$1/*/*.db 2>/dev/null | head -1; }
python scripts/kernel_roofline.py $(db rf_trace) $(db rf_FETCH_SIZE) $(db rf_WRITE_SIZE) $(db rf_sq) > gpurun_out/r06_kernel_roofline.txt 2> gpurun_out/rf_err.log
rm -rf gpurun_out/rf_trace gpurun_out/rf_FETCH_SIZE gpurun_out/rf_WRITE_SIZE gpurun_out/rf_sq
PMC_TAG=r06 bash scripts/gpu_pmc_traffic.sh > gpurun_out/z_pmc_traffic.log 2>&1
cp profiles/r06_pmc_traffic.json profiles/r06_pmc_traffic_per_kernel.txt gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
tail -4 gpurun_out/z_test_full.log; cat gpurun_out/z_smoke.log; cat gpurun_out/z_two_stream_identity.log
python - <<'P'
import json
for f in ("z_bench_cfg2", "z_bench_torchrun"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.log").read().strip().splitlines()[-1])
        r=d.get('roofline',{})
        print(f, d['value'], d['ms_per_step'], "conv", r.get('frac'), (r.get('executed') or {}).get('frac'), "traffic", r.get('traffic'), (r.get('traffic_source') or {}).get('live'), "score", d.get('score_gemm',{}).get('frac'), d.get('score_gemm',{}).get('us_per_step'), "hbm", d.get('hbm_family',{}).get('ms_per_step'), d.get('schedules'))
        for k,v in d.get('also',{}).items(): print("   ", k, {a:b for a,b in v.items() if a in ('value','ms_per_step','error','vs_engine_path','two_stream_speedup','score_path')}, (v.get('score_gemm') or {}).get('frac'))
        if 'cpu_baseline' in d: print("   cpu", d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.log").read()[-300:])
P
head -40 gpurun_out/r06_kernel_roofline.txt
