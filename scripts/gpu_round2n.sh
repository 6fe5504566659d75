#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv_fwd or conv_dgrad" 2>&1 | tail -3) > gpurun_out/n_tests.log
(timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_engine_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3) >> gpurun_out/n_tests.log
for c in cfg4 cfg5; do
  (timeout 300 python bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/n_bench_$c.log
  (DPC_IGEMM_WS_TGROUP=0 timeout 300 python bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1) > gpurun_out/n_bench_${c}_off.log
done
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/n_bench_cfg2.log
cat gpurun_out/n_tests.log; for f in n_bench_cfg4 n_bench_cfg4_off n_bench_cfg5 n_bench_cfg5_off n_bench_cfg2; do python -c "import json,sys; d=json.loads(open('gpurun_out/$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
