#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_data_pipeline.py -m gpu -q -p no:cacheprovider -k "dgrad or frames or engine_consumes" 2>&1 | tail -3) > gpurun_out/f_tests.log
(timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_configs_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3) >> gpurun_out/f_tests.log
(timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/f_bench_cfg2.log
(timeout 300 python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1) > gpurun_out/f_bench_cfg2_b.log
(timeout 300 python bench.py --config cfg4 --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/f_bench_cfg4.log
cat gpurun_out/f_tests.log; for f in f_bench_cfg2 f_bench_cfg2_b f_bench_cfg4; do python -c "import json,sys; d=json.loads(open('gpurun_out/$f.log').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'])"; done
