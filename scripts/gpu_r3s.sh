#!/bin/bash
# round 3, session s: tiled weight repack, XCD-paired column tiles of the generic kernel
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/s_test_kernels.log
(REPS=20 timeout 200 python scripts/head_bench.py 2>&1 | $F | grep -i "score GEMM\|d_pred split\|d_finf w\|CE") > gpurun_out/s_head_bench.txt
B="python bench.py --no-cpu-baseline --no-also --steps 100"
(timeout 300 $B 2>&1 | tail -1) > gpurun_out/s_bench_cfg2.log
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s_prof -o s -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -2) > $R/gpurun_out/s_rocprof.log
cd $R
python scripts/rocpd_stats.py gpurun_out/s_prof/*.db > gpurun_out/s_kernel_stats.txt 2>&1 || ls gpurun_out/s_prof
rm -rf gpurun_out/s_prof
cat gpurun_out/s_test_kernels.log gpurun_out/s_head_bench.txt; head -45 gpurun_out/s_kernel_stats.txt | cut -c1-150
python - <<'P'
import json
d=json.loads(open('gpurun_out/s_bench_cfg2.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['score_gemm'])
P
