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


DEFAULT_CHANNELS = 8   # RCCL channel workgroups (one CU each) of the gradient all-reduce, and CUs the persistent kernels leave to them


def default_reserve_cus(world: int, environ=os.environ) -> int:
    """CUs the one-workgroup-per-CU kernels leave free while the tail all-reduce is in flight (DPCEngine.reserve_cus):
    DPC_RESERVE_CUS when set, else the RCCL channel count when there is more than one rank, else 0."""
    if environ.get("DPC_RESERVE_CUS") is not None:
        return int(environ["DPC_RESERVE_CUS"])
    if world <= 1:
        return 0
    return int(environ.get("DPC_RCCL_CHANNELS") or environ.get("NCCL_MAX_NCHANNELS") or DEFAULT_CHANNELS)


def configure_rccl(world: int = 1, environ=os.environ) -> dict:
    """RCCL environment for the gradient exchange; call BEFORE the first HIP call of the process (RCCL reads its environment when
    the communicator is created, the HSA runtime when it starts).  Returns what was set (bench.py prints it in the JSON line).

    Why few channels.  The tail all-reduce (58 / 132 MB, dpc/main.py:65's DataParallel reduce) runs under ~9 ms of layer1 + stem
    backward: it needs ~7 GB/s to hide completely, so bandwidth is not what to optimise -- the CUs it takes from the backward
    pass are.  Measured with a co-tenant of k workgroups x 64 KB LDS for 1 ms on one MI355X (scripts/probes/cotenant_step.py,
    profiles/r04_cotenant.txt): overlapped with all CUs claimed by the persistent grids +1.7..1.8 % of a step, with k CUs left
    free +0.7 % (k = 8) / +0.8 % (16) / +1.3 % (32), not overlapped at all +4.5 %.  Hence for world > 1:
      NCCL_MAX_NCHANNELS = 8 unless it is set explicitly (DPC_RCCL_CHANNELS=n pins NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = n),
      DPC_RESERVE_CUS defaults to the same number (default_reserve_cus; 0 switches the carve-out off).
    HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC is what the host driver supports) is exported for child processes if the launcher
    dropped it; in THIS process it only takes effect when the HSA runtime has not started yet, and is reported accordingly."""
    applied = {}
    if environ.get("DPC_RCCL_CHANNELS"):       # asked for: pin the channel count
        ch = int(environ["DPC_RCCL_CHANNELS"])
        for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS"):
            environ[k] = str(ch)
            applied[k] = ch
    elif world > 1 and "NCCL_MAX_NCHANNELS" not in environ:   # default: only a cap (a forced minimum could exceed what a topology offers)
        environ["NCCL_MAX_NCHANNELS"] = str(DEFAULT_CHANNELS)
        applied["NCCL_MAX_NCHANNELS"] = DEFAULT_CHANNELS
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in environ:
        environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        started = torch.cuda.is_initialized()
        applied["HSA_ENABLE_IPC_MODE_LEGACY"] = "0 (child processes only: the HIP runtime of this process had already started)" if started else 0
    applied["DPC_RESERVE_CUS"] = default_reserve_cus(world, environ)
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


def gather_step_times(dist, seconds: float, steps: int, device, world: int):
    """bench.py's clock: every rank timed the same K steps between barriers; returns (max over ranks in seconds, [ms per step of every
    rank]).  The max is what `value` is computed from; the list shows stragglers as a spread."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    if dist is None:
        return seconds, [1e3 * seconds / steps]
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item(), [1e3 * e.item() / steps for e in every]


def ranks_agree(dist, ok: bool, device) -> bool:
    """every rank learns whether ALL ranks got here without an exception (one tiny all-reduce, same place on every rank): what one
    rank could not do (a capture refused, out of memory), no rank goes on with -- their collectives would no longer pair up"""
    if dist is None:
        return bool(ok)
    f = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return bool(f.item() > 0.5)
