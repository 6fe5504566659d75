cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|^\S*SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+\[?|GRBM_[A-Z_]+|TCP_[A-Z_0-9]+" | sort -u | head -400 > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  (timeout 300 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- $B 2>&1 | tail -3) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log
done
ls -R $GRAFT_REPO_ROOT/gpurun_out | head -40 > $GRAFT_REPO_ROOT/gpurun_out/ls.txt
