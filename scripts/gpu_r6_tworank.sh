#!/bin/bash
# Two ranks of bench.py on the ONE GPU of a gpurun box (DPC_BENCH_REHEARSAL=1: both on cuda:0, gloo carrying the collectives): the
# control flow the driver launches on an 8-GPU node -- torch.distributed.run, barriers, three-graph replay with a live exchange
# between the graphs, the clock's all_gather / MAX, the side schedules with their rank agreement -- executed end to end.  RCCL itself
# (refuses two ranks on one device) is the one thing this cannot run.  Output: gpurun_out/r06_tworank_rehearsal.txt
mkdir -p gpurun_out
out=gpurun_out/r06_tworank_rehearsal.txt
: > $out
export DPC_BENCH_REHEARSAL=1
for args in "--batch 32 --steps 6 --warmup 2 --pmc off --schedules --no-cpu-baseline --no-also" "--batch 32 --steps 6 --warmup 2 --pmc off --no-graph --no-cpu-baseline --no-also" "--steps 5 --warmup 2"; do
  echo "== torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 $args" >> $out
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 $args >> $out 2> gpurun_out/tworank_err.txt
  echo "rc=$?" >> $out
  tail -5 gpurun_out/tworank_err.txt >> $out
done
# the same engine, one rank, same per-rank batch: what each rank of the rehearsal computes alone (loss of rank 0 must agree at step 1 only:
# after that the averaged gradients differ by construction)
unset DPC_BENCH_REHEARSAL
echo "== one rank, batch 32" >> $out
timeout 300 python bench.py --config cfg2 --batch 32 --steps 6 --warmup 2 --pmc off --no-roofline >> $out 2>> gpurun_out/tworank_err.txt
echo "rc=$?" >> $out
tail -c 3000 $out
