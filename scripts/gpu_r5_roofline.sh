#!/bin/bash
# round 5: per-kernel HBM GB/s and MFMA utilisation of the cfg2 step (one stream, kernel by kernel): four rocprofv3 passes -> profiles/r05_kernel_roofline.txt
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-also --no-graph --pmc off"
export DPC_WGRAD_STREAM=0
(timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/rf_trace -o t -- $B 2>&1 | tail -1) > $R/gpurun_out/rf.log
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/rf_$c -o t -- $B 2>&1 | tail -1) >> $R/gpurun_out/rf.log
done
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/rf_sq -o t -- $B 2>&1 | tail -1) >> $R/gpurun_out/rf.log
cd $R
db() { ls gpurun_out/$1/*.db gpurun_out/$1/*/*.db 2>/dev/null | head -1; }
python scripts/kernel_roofline.py $(db rf_trace) $(db rf_FETCH_SIZE) $(db rf_WRITE_SIZE) $(db rf_sq) > gpurun_out/r05_kernel_roofline.txt 2> gpurun_out/rf_err.log
rm -rf gpurun_out/rf_trace gpurun_out/rf_FETCH_SIZE gpurun_out/rf_WRITE_SIZE gpurun_out/rf_sq
cat gpurun_out/rf.log gpurun_out/rf_err.log; head -50 gpurun_out/r05_kernel_roofline.txt
