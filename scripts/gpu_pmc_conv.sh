#!/bin/bash
# SQ counter tables (three passes) of one cfg2 train step: MFMA busy, LDS conflicts, wait buckets, and the shader clock
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/cq_$i -o pmc -- $B 2>&1 | tail -2) > $R/gpurun_out/cq_$i.log
  f=$(ls $R/gpurun_out/cq_$i/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_table.py $f "." > $R/gpurun_out/cq_$i.txt 2>&1
  [ -n "$f" ] && [ $i = 3 ] && python $R/scripts/rocpd_stats.py $f > $R/gpurun_out/cq_3_stats.txt 2>&1
  rm -rf $R/gpurun_out/cq_$i/*.db
done
head -12 $R/gpurun_out/cq_1.txt | cut -c1-220
