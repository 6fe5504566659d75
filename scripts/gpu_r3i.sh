#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 200 python scripts/stem_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/i_stem_bench.txt
(for d in 0 4 36 8 60 127; do echo "DPC_SF_DBG=$d"; DPC_SF_DBG=$d timeout 100 python scripts/stem_bench.py --probe --iters 5 2>&1 | grep -v amdgpu.ids | grep fused; done) > gpurun_out/i_stem_probe.txt
(timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/i_bench.log
(DPC_STEM_FUSED=0 timeout 300 python bench.py --no-cpu-baseline --no-also 2>&1 | tail -1) > gpurun_out/i_bench_twokernel.log
cat gpurun_out/i_stem_bench.txt gpurun_out/i_stem_probe.txt
for f in i_bench i_bench_twokernel; do python -c "
import json; d=json.loads(open('gpurun_out/$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['hbm_family']['ms_per_step'], d['wgrad_kernel'], d['score_gemm']['frac'])"; done
