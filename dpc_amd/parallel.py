"""Data parallelism for the DPC-RNN step: one process per GPU, one exchange per step.

The reference wraps the model in nn.DataParallel (dpc/main.py:65): single process, a thread
per GPU, parameters re-broadcast every forward, score+mask gathered to GPU0, gradients
reduce-added on GPU0.  Its own comments note that the similarity matrix, the BN statistics
and the loss are per-GPU (dpc/main.py:180,211-213), so the MI355X equivalent is plain DDP
semantics: every rank runs the engine on its shard and the flat f32 gradient arena
(58.3 MB r18 / 131.8 MB r34) is averaged with ONE RCCL all-reduce over xGMI -- no gather,
no parameter broadcast.  (mean over all rows == mean of equal-sized per-rank means.)
"""
from __future__ import annotations

import torch


def make_allreduce(dist, world: int, force: bool = False):
    """returns f(flat_grad) that averages the arena across ranks in place (backend 'nccl' == RCCL)"""
    if dist is None or (world <= 1 and not force):
        return None
    backend = dist.get_backend()

    def allreduce(flat: torch.Tensor):
        if backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:  # gloo (CPU tests): no AVG
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(world)
    return allreduce


def shard_of(global_batch: int, world: int, rank: int) -> slice:
    """contiguous dim-0 shard of the global batch owned by `rank` (drop_last semantics, main.py:313)"""
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)
