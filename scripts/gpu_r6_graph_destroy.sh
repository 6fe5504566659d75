#!/bin/bash
# round 6: graph destruction switched back ON (DPC_KEEP_GRAPHS=0) under a guarded host allocator -- glibc's
# MALLOC_CHECK_=3 (abort on heap corruption at the next malloc / free that sees it), MALLOC_PERTURB_ (freed memory is overwritten: a
# use-after-free reads garbage instead of the old value) -- on the first pytest run of a fresh box, the situation that produced the
# corruption in rounds 3-4.  ARM=A: captured events released when side() returns (rounds 3-4's behaviour), host kernel arguments
# (HIP_FORCE_DEV_KERNARG=0: what every failing run had); ARM=B: the same with the events kept until their graphs are gone.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
ev=0; [ "${ARM:-A}" = B ] && ev=1
MALLOC_CHECK_=3 MALLOC_PERTURB_=165 HIP_FORCE_DEV_KERNARG=0 DPC_KEEP_GRAPHS=0 DPC_KEEP_CAPTURE_EVENTS=$ev timeout 700 python -X faulthandler -m pytest tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/gd6_last.log 2>&1
rc=$?
echo "arm ${ARM:-A} (events kept: $ev, MALLOC_CHECK_=3 MALLOC_PERTURB_=165, host kernarg) rc=$rc $(tail -1 gpurun_out/gd6_last.log | tr -d '=' | cut -c1-100) $(grep -m1 -E 'KeyError|Segmentation|Fatal Python|core dumped|Aborted|corrupt|double free|invalid pointer' gpurun_out/gd6_last.log | cut -c1-160)" | tee gpurun_out/gd6_line.txt
