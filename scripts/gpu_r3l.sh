#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 200 python scripts/stem_bench.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/l_stem_bench.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  (timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/lsq_$i -o pmc -- python $R/scripts/stem_bench.py --iters 2 2>&1 | tail -2) > $R/gpurun_out/lsq_$i.log
  f=$(ls $R/gpurun_out/lsq_$i/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_table.py $f "stem" > $R/gpurun_out/l_sq_$i.txt 2>&1
  rm -rf $R/gpurun_out/lsq_$i
done
cd $R; cat gpurun_out/l_stem_bench.txt gpurun_out/l_sq_*.txt
