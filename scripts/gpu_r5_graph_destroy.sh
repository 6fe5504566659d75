#!/bin/bash
# round 5: does the host-memory corruption after graph destruction (rounds 3-4: 4 of 12 first runs on a fresh box) come from captured
# events that die before their graph?  One fresh box per call (run this script once per `gpurun` call), first pytest run of the two
# files that showed it, graphs DESTROYED with their engines (DPC_KEEP_GRAPHS=0); ARM=A: fork / join events released when side()
# returns (rounds 3-4), ARM=B: kept alive until after their graphs are gone.  Appends one line to gpurun_out/r05_graph_destroy.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
# ARM=C: arm A with kernel arguments back in host memory (HIP_FORCE_DEV_KERNARG=0: what every run of rounds 3-4 that failed had).
ev=0; [ "${ARM:-A}" = B ] && ev=1
ka=1; [ "${ARM:-A}" = C ] && ka=0
HIP_FORCE_DEV_KERNARG=$ka DPC_KEEP_GRAPHS=0 DPC_KEEP_CAPTURE_EVENTS=$ev timeout 500 python -X faulthandler -m pytest tests/test_engine_gpu.py tests/test_graph_rccl_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/gd_last.log 2>&1
rc=$?
echo "arm ${ARM:-A} (events kept: $ev, dev kernarg: $ka) rc=$rc $(tail -1 gpurun_out/gd_last.log | tr -d '=' | cut -c1-100) $(grep -m1 -E 'KeyError|Segmentation|Fatal Python|core dumped|Aborted' gpurun_out/gd_last.log | cut -c1-160)" | tee gpurun_out/gd_line.txt
