#!/bin/bash
# SQ counters of the stem kernels (plain vs fused) from scripts/stem_bench.py
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/ksq_$i -o pmc -- python $R/scripts/stem_bench.py --iters 2 2>&1 | tail -2) > $R/gpurun_out/ksq_$i.log
  f=$(ls $R/gpurun_out/ksq_$i/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_table.py $f "stem|pool" > $R/gpurun_out/k_sq_$i.txt 2>&1
  rm -rf $R/gpurun_out/ksq_$i
done
cat $R/gpurun_out/k_sq_*.txt
