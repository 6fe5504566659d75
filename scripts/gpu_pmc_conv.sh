#!/bin/bash
# PMC counters for one conv micro-benchmark: PMC_ARGS="l2 --ops fwd --iters 3"
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/pmcc_*
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcc_$i -o pmc -- python $R/scripts/conv_bench.py $PMC_ARGS 2>&1 | tail -2) > $R/gpurun_out/pmcc_$i.log
  python $R/scripts/pmc_table.py $R/gpurun_out/pmcc_$i/*.db "${PMC_PAT:-igemm|wgrad|halo}"
done
