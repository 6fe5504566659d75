#!/bin/bash
# round-3 final session: gpu tier, smoke, bench lines, profiles, PMC traffic
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 1700 python -m pytest tests -m gpu -q -rP -p no:cacheprovider 2>&1) > gpurun_out/z_test_full.log
grep -E "passed|failed|error|bf16 anchor:|cfg5 fused|LC gradients|fold_c1|Error|assert " gpurun_out/z_test_full.log | tail -60 > gpurun_out/z_test.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > gpurun_out/z_smoke.log
(timeout 900 python bench.py 2>&1 | tail -2) > gpurun_out/z_bench_cfg2.log
(DPC_FOLD=0 DPC_STEM_FUSED=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-also 2>&1 | tail -2) > gpurun_out/z_bench_cfg2_r2paths.log
(timeout 300 python bench.py --score-path fused --steps 100 --no-cpu-baseline --no-also 2>&1 | tail -2) > gpurun_out/z_bench_cfg2_fused.log
(DPC_WGRAD_STREAM=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/z_bench_cfg2_one_stream.log
(DPC_IGEMM_WS_PAR=0 DPC_PARITY_ILV=0 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-also 2>&1 | tail -2) > gpurun_out/z_bench_cfg2_old_dgrad.log
(timeout 400 python bench.py --dtype f32 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/z_bench_cfg2_f32.log
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -2) > gpurun_out/z_bench_torchrun.log
(DPC_RESERVE_CUS=16 DPC_RCCL_CHANNELS=4 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 100 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -2) > gpurun_out/z_bench_torchrun_knobs.log
(timeout 300 python -m dpc_amd.main --net resnet18 --img_dim 128 --batch_size 32 --gpu 0 --epochs 1 --synthetic 6 --print_freq 2 --save_dir gpurun_out/ck_dpc 2>&1 | tail -8) > gpurun_out/z_main.log
(timeout 300 python -m dpc_amd.lc_main --net resnet18 --img_dim 128 --batch_size 16 --gpu 0 --epochs 2 --synthetic 6 --print_freq 2 --pretrain gpurun_out/ck_dpc/epoch1.pth.tar --save_dir gpurun_out/ck_lc 2>&1 | tail -12) > gpurun_out/z_lc_main.log
ls gpurun_out/ck_lc >> gpurun_out/z_lc_main.log 2>&1
rm -rf gpurun_out/ck_dpc gpurun_out/ck_lc
(timeout 200 python scripts/stem_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/z_stem_bench.txt
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/z_head_bench.txt
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03 -o bench -- python $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -3) > $R/gpurun_out/z_rocprof.log
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03_cfg4 -o bench -- python $R/bench.py --config cfg4 --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/z_rocprof4.log
(timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03_cfg5 -o bench -- python $R/bench.py --config cfg5 --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3) > $R/gpurun_out/z_rocprof5.log
cd $R
for d in prof_r03 prof_r03_cfg4 prof_r03_cfg5; do
  f=$(ls gpurun_out/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocpd_stats.py $f > gpurun_out/${d}_stats.txt 2>&1
done
f=$(ls gpurun_out/prof_r03/*.db 2>/dev/null | head -1); [ -n "$f" ] && python scripts/timeline.py $f "." > gpurun_out/prof_r03_timeline.txt 2>&1
rm -rf gpurun_out/prof_r03/*.db gpurun_out/prof_r03_cfg4/*.db gpurun_out/prof_r03_cfg5/*.db
PMC_TAG=r03 bash scripts/gpu_pmc_traffic.sh > gpurun_out/z_pmc_traffic.log 2>&1
cp profiles/r03_pmc_traffic.json profiles/r03_pmc_traffic_per_kernel.txt gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_FETCH_SIZE/*.db gpurun_out/pmc_WRITE_SIZE/*.db
cat gpurun_out/z_test.log | tail -12; cat gpurun_out/z_smoke.log; for f in gpurun_out/z_bench_*.log; do echo "$f: $(tail -1 $f | cut -c1-160)"; done
