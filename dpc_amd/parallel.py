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

import os

import torch


def configure_rccl(environ=os.environ) -> dict:
    """Knobs for the first multi-GPU runs, applied BEFORE init_process_group (RCCL reads its environment when the communicator
    is created).  None is set unless asked for; what is set is returned (bench.py prints it in the JSON line).
      DPC_RCCL_CHANNELS=n   NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = n: the number of channel workgroups (one CU each) an
                            all-reduce occupies.  The 58 / 132 MB gradient over 7 xGMI links wants few, long-lived channels next to
                            the backward kernels; RCCL's default is sized for a collective that owns the chip.
      DPC_RESERVE_CUS=n     (read by DPCEngine) the persistent one-workgroup-per-CU kernels of layer1 + stem backward launch
                            256 - n workgroups while the tail all-reduce is in flight, so the channel kernels are not queued
                            behind them.  A/B both against the defaults (0 / unset) on the first 8-GPU run: DESIGN.md section 7.
      HSA_ENABLE_IPC_MODE_LEGACY=0 is kept (dmabuf IPC is what the host driver supports); set here if the launcher dropped it."""
    applied = {}
    ch = environ.get("DPC_RCCL_CHANNELS")
    if ch:
        for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS"):
            environ[k] = str(int(ch))
            applied[k] = int(ch)
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in environ:
        environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        applied["HSA_ENABLE_IPC_MODE_LEGACY"] = 0
    if environ.get("DPC_RESERVE_CUS"):
        applied["DPC_RESERVE_CUS"] = int(environ["DPC_RESERVE_CUS"])
    return applied


class GradAllReduce:
    """Averages the flat gradient arena across ranks in place (backend 'nccl' == RCCL over xGMI).

    Two buckets so that the exchange hides behind the backward pass: the arena is laid out in forward order
    (stem, layer1, ..., layer4, ConvGRU, predictor) and the backward pass finishes it back to front, so the
    tail [split:] -- everything except the stem and layer1, 99 % of the bytes -- is final while layer1 and the
    stem (the HBM-heavy ~10 ms of the step at r18/128/B=128) are still running.  `start(tail)` launches that
    all-reduce asynchronously (RCCL runs it on its own stream, ordered after the work already queued);
    `finish(head)` reduces the small head and joins.  Calling the object reduces the whole arena at once.
    """

    def __init__(self, dist, world: int):
        self.dist, self.world = dist, world
        self.avg = dist.get_backend() == "nccl"  # gloo (CPU tests) has no AVG
        self._pending = []

    def _launch(self, t: torch.Tensor, async_op: bool):
        op = self.dist.ReduceOp.AVG if self.avg else self.dist.ReduceOp.SUM
        return self.dist.all_reduce(t, op=op, async_op=async_op)

    def start(self, tail: torch.Tensor):
        self._pending.append((self._launch(tail, True), tail))

    def finish(self, head: torch.Tensor):
        if head.numel():
            self._launch(head, False)
            if not self.avg:
                head.div_(self.world)
        for work, t in self._pending:
            work.wait()
            if not self.avg:
                t.div_(self.world)
        self._pending.clear()

    def __call__(self, flat: torch.Tensor):
        self.finish(flat)


def make_allreduce(dist, world: int, force: bool = False):
    """returns a GradAllReduce (callable on the whole arena, or start()/finish() on its two buckets), or None"""
    if dist is None or (world <= 1 and not force):
        return None
    return GradAllReduce(dist, world)


def shard_of(global_batch: int, world: int, rank: int) -> slice:
    """contiguous dim-0 shard of the global batch owned by `rank` (drop_last semantics, main.py:313)"""
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)
