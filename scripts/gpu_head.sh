#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "score_fused or gru_chain" 2>&1 | tail -5) > gpurun_out/h_kernels.log
(timeout 300 python scripts/head_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/h_bench.log
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  (REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/hpmc_$tag -o pmc -- python $R/scripts/head_bench.py 2>&1 | tail -2) > $R/gpurun_out/hpmc_$tag.log
  f=$(ls $R/gpurun_out/hpmc_$tag/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_table.py $f "score|gru_chain" > $R/gpurun_out/hpmc_$tag.txt 2>&1
  rm -rf $R/gpurun_out/hpmc_$tag/*.db
done
cat $R/gpurun_out/h_bench.log
