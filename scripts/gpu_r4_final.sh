#!/bin/bash
# HISTORICAL (round 4): sets DPC_SCORE_GEMM_COUNTED / DPC_EARLY_FINALIZE, switches that were removed in round 5 -- those A/B arms now run
# identical code.  Kept for the record of what round 4 measured; the current recipes are scripts/gpu_r6_*.sh.
# round 4 closing session: GPU tier, head timings and one bench line on the final build, PMC traffic re-measured on the final kernel sources
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/g_test_full.log 2>&1; echo "rc=$?" >> gpurun_out/g_test_full.log
(DPC_SCORE_GEMM_COUNTED=0 timeout 300 python scripts/head_bench.py 2>&1 | grep "score GEMM") > gpurun_out/g_head_uncounted.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/g_head.log
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/g_bench_cfg2.log
PMC_TAG=r04 bash scripts/gpu_pmc_traffic.sh > gpurun_out/g_pmc_traffic.log 2>&1
cp profiles/r04_pmc_traffic.json profiles/r04_pmc_traffic_per_kernel.txt gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
tail -4 gpurun_out/g_test_full.log; cat gpurun_out/g_head_uncounted.log; grep "score GEMM" gpurun_out/g_head.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/g_bench_cfg2.log").read().strip().splitlines()[-1])
print("cfg2", d['value'], d['ms_per_step'], "conv", d['roofline']['frac'], "traffic", d['roofline']['traffic'], d['roofline']['traffic_source'], "score", d['score_gemm']['frac'], d['score_gemm']['us_per_step'])
for k,v in d.get('also',{}).items(): print("   ", k, {a:b for a,b in v.items() if a in ('value','ms_per_step','error','vs_engine_path','two_stream_speedup')})
P
