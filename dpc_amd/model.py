"""Drop-in module boundary: ``DPC_RNN(sample_size, num_seq, seq_len, pred_step, network)``
with ``forward(block) -> [score, mask]`` exactly as dpc/model_3d.py:14-98, backed by the
MI355X engine.  Same constructor arguments, public attributes (.backbone/.agg/
.network_pred/.mask/.param/.last_size/.last_duration/.pred_step), state_dict keys
(including the duplicated ``agg.cell_list.0.*`` alias, backbone/convrnn.py:55-58) and
error behaviour (``IOError('model type is wrong')``, backbone/select_backbone.py:19).

Parameters are ``nn.Parameter`` views into the engine's flat f32 arena, so
``load_state_dict`` / ``state_dict`` / any torch optimizer see the reference layout while
the kernels (and the fused Adam of ``engine.train_step``) work on one contiguous buffer.
``forward`` returns a score that autograd can differentiate: its backward runs the
engine's hand-written backward kernels and hands the parameter gradients to autograd.
There is no CPU path: calling it with CPU tensors raises (dpc_amd._lib.DpcError).

Nothing is copied at the boundary (round 4): the returned score is the engine's own buffer
(two alternate, so the previous step's score outlives one more forward), the gradients
reach autograd as fresh views of the gradient arena, which AccumulateGrad adopts as
``.grad``, and ``dpc_amd.optim.Adam`` is the fused arena update behind
``torch.optim.Optimizer``'s interface.  ``model.bfloat16()`` / ``model.float()`` (or
``DPC_COMPUTE_DTYPE=bf16``) select the kernels' operand type; parameters stay f32.
bench.py times the reference's own loop lines over this module (``also.module``).
"""
from __future__ import annotations

import math
import os
import threading
import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .engine import DPCEngine, LAYER_PLAN, LAYER_WIDTH, param_shapes


_DTYPE_NAMES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "float32": torch.float32}

# nn.DataParallel over several devices (dpc/main.py:65-66 with more than one id in --gpu): torch replicates the module per forward --
# shallow copies whose parameters are broadcast non-leaf tensors (torch nn/parallel/replicate.py) -- and runs them in one thread per
# device (data_parallel.py:173-198).  The engine behind this module is ONE device's static schedule (buffers, arenas), so a replica
# does not own one: it looks up the engine of ITS device in a registry that hangs off the root module (built at the replica's first
# forward there, kept across steps), copies the broadcast parameters into that engine's arena -- the 58 MB per device and step that
# the reference's broadcast costs as well -- and returns a score whose backward hands the engine's gradients to the broadcast
# copies, so torch's Broadcast.backward reduce-adds them onto the root's parameters on device 0 exactly as for the reference.
# Negatives, BatchNorm statistics and dropout are per replica, as they are under the reference's DataParallel (dpc/main.py:180,
# 211-213).  This is the reference's unmodified entry made to work, not the fast route: one process per GPU with an RCCL all-reduce
# (`python -m dpc_amd.main --gpu 0,1,...`, dpc_amd/parallel.py) keeps parameters resident, overlaps the exchange and replays graphs.
_REPLICA_NOTE = ("dpc_amd.DPC_RNN under nn.DataParallel over several devices: every forward re-broadcasts the parameters and copies them "
                 "into a per-device engine (the reference's DataParallel semantics, kept for drop-in compatibility).  The fast multi-GPU "
                 "route is one process per GPU with an RCCL gradient all-reduce: `python -m dpc_amd.main --gpu 0,1,...` (same flags as "
                 "dpc/main.py) or torch.nn.parallel.DistributedDataParallel under torch.distributed.run")


# the downstream classifier (dpc_amd/lc.py) keeps the single-device rule
_REPLICA_ERROR = ("dpc_amd.DPC_RNN cannot run as an nn.DataParallel replica (DataParallel over more than one device replicates the "
                  "module every forward; the MI355X engine is a single-device static schedule).  Multi-GPU training is one process "
                  "per GPU with an RCCL gradient all-reduce: `python -m dpc_amd.main --gpu 0,1,...` (same flags as dpc/main.py), or "
                  "wrap the module in torch.nn.parallel.DistributedDataParallel under torch.distributed.run; with a single device "
                  "nn.DataParallel(model, device_ids=[0]) works unchanged")


class _Holder(nn.Module):
    """namespace module so that parameters appear under the reference's dotted names"""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            setattr(mod, p, _Holder())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], param)


def _init_reference_style(shapes: Dict[str, tuple], gen: torch.Generator) -> Dict[str, torch.Tensor]:
    """kaiming_normal(fan_out) convs + BN 1/0 (backbone/resnet_2d3d.py:224-230); orthogonal gain 1,
    zero bias for agg / network_pred (dpc/model_3d.py:100-106)."""
    out = {}
    for k, shp in shapes.items():
        if k.startswith("backbone") and len(shp) == 5:
            fan_out = shp[0] * shp[2] * shp[3] * shp[4]
            out[k] = torch.randn(shp, generator=gen) * math.sqrt(2.0 / fan_out)
        elif k.startswith("backbone"):
            out[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        elif k.endswith("bias"):
            out[k] = torch.zeros(shp)
        else:
            w = torch.empty(shp)
            nn.init.orthogonal_(w, 1, generator=gen)
            out[k] = w
    return out


class _DPCScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, block, train, masks, *params):
        eng = model._engine
        # The engine keeps ONE set of saved activations (static schedule): the graph node is only valid until the
        # next forward through the same engine, and for one backward.
        model._fwd_generation += 1
        # The returned score IS the engine's buffer (no 151 MB / 983 MB copy per forward).  Two buffers alternate, so the score of
        # the previous forward stays readable for one more step (the reference's loop deletes it before the next iteration,
        # dpc/main.py:227); after that the handle shows a later step's values.
        eng.score = model._score_buffers(eng)[model._fwd_generation & 1]
        score = eng.forward(block, train=train, dropout_masks=masks)
        ctx.model = model
        ctx.generation = model._fwd_generation
        ctx.used = False
        return score

    @staticmethod
    def backward(ctx, dscore):
        model = ctx.model
        eng = model._engine
        if ctx.generation != model._fwd_generation:
            raise RuntimeError("dpc_amd.DPC_RNN: backward of a score whose saved activations were overwritten by a later "
                               "forward (the engine holds one step's activations; call backward before the next forward)")
        if ctx.used:
            raise RuntimeError("dpc_amd.DPC_RNN: trying to backward through the graph a second time (activations are "
                               "consumed in place by the backward kernels)")
        ctx.used = True
        names = model._param_names
        # Parameter gradients go back through autograd as FRESH views of the engine's gradient arena: AccumulateGrad adopts a
        # gradient nobody else holds instead of copying it, so after loss.backward() a parameter's .grad IS its slice of the arena
        # (no 58 MB of copies; dpc_amd.optim.Adam and torch's foreach Adam read it in place) while everything that hooks the
        # autograd graph -- DistributedDataParallel's reducer, parameter hooks -- still sees every gradient arrive.
        named = dict(model.named_parameters())
        aliased = any(named[k].grad is not None and named[k].grad.data_ptr() == eng.G[k].data_ptr() for k in names)
        old = eng.flat_g.clone() if aliased else None   # .grad still set (no zero_grad): torch accumulates; the kernels overwrite the arena
        eng.backward(dscore_external=dscore)
        src = eng.flat_g
        if old is not None:   # hand autograd a copy of the new gradients and give the arena (= the live .grad tensors) its old values back
            src = eng.flat_g.clone()
            eng.flat_g.copy_(old)
        grads = []
        for k in names:
            o, n = eng.offsets[k]
            grads.append(src[o:o + n].view(eng.shapes[k]))
        return (None, None, None, None) + tuple(grads)


class _DPCScoreReplica(torch.autograd.Function):
    """the score of one nn.DataParallel replica: forward on the replica's device engine, backward hands that engine's gradients to the
    broadcast parameter copies (views of its gradient arena: Broadcast.backward's reduce-add reads them before the next forward)"""

    @staticmethod
    def forward(ctx, eng, names, block, train, masks, *params):
        eng.load_params({k: p.detach() for k, p in zip(names, params)})
        eng._replica_generation = getattr(eng, "_replica_generation", 0) + 1
        score = eng.forward(block, train=train, dropout_masks=masks)
        ctx.eng, ctx.names, ctx.generation, ctx.used = eng, names, eng._replica_generation, False
        return score

    @staticmethod
    def backward(ctx, dscore):
        eng = ctx.eng
        if ctx.generation != eng._replica_generation or ctx.used:
            raise RuntimeError("dpc_amd.DPC_RNN (DataParallel replica): backward of a score whose saved activations are gone (a later "
                               "forward on the same device, or a second backward)")
        ctx.used = True
        eng.backward(dscore_external=dscore)
        return (None, None, None, None, None) + tuple(eng.G[k] for k in ctx.names)


class DPC_RNN(nn.Module):
    """DPC with RNN (dpc/model_3d.py:14)"""

    def __init__(self, sample_size, num_seq=8, seq_len=5, pred_step=3, network="resnet50",
                 compute_dtype=torch.float32, widths=LAYER_WIDTH, seed: int = 0, _simulator: Optional[L.Lib] = None):
        super().__init__()
        if network not in LAYER_PLAN:
            raise IOError("model type is wrong")  # select_backbone.py:19 (resnet50+ are outside this build's scope)
        self.sample_size, self.num_seq, self.seq_len, self.pred_step = sample_size, num_seq, seq_len, pred_step
        self.network = network
        self.last_duration = int(math.ceil(seq_len / 4))
        self.last_size = int(math.ceil(sample_size / 32))
        self.param = {"feature_size": widths[3], "num_layers": 1, "hidden_size": widths[3]}
        self.compute_dtype = compute_dtype
        self.widths = tuple(widths)
        self.mask = None
        self._engine: Optional[DPCEngine] = None
        self._engine_key = None
        self._param_names: List[str] = []
        self._forced_masks = None
        self._fwd_generation = 0
        self._simulator = _simulator  # tests only: the host-side SIMT simulator handle (CPU tier); never set by the product
        self._score_bufs = None
        # engines of nn.DataParallel replicas by (device, slot, batch, compute dtype); the dict and its lock are shared with every replica
        # (a replica's __dict__ is a shallow copy of the root's: torch nn/modules/module.py _replicate_for_data_parallel)
        self._replica_engines: Dict[tuple, DPCEngine] = {}
        self._replica_lock = threading.Lock()
        if os.environ.get("DPC_COMPUTE_DTYPE"):   # select the throughput mode without touching the reference's constructor call
            want = os.environ["DPC_COMPUTE_DTYPE"].lower()
            if want not in _DTYPE_NAMES:
                raise ValueError(f"DPC_COMPUTE_DTYPE={os.environ['DPC_COMPUTE_DTYPE']!r}: accepted values are {sorted(_DTYPE_NAMES)}")
            self.compute_dtype = _DTYPE_NAMES[want]
        shapes = param_shapes(network, widths)
        init = _init_reference_style(shapes, torch.Generator().manual_seed(seed))
        for k, shp in shapes.items():
            _attach(self, k, nn.Parameter(init[k]))
            self._param_names.append(k)
        # the reference registers the single GRU cell twice (convrnn.py:55-58): alias module, same Parameters
        self.agg.cell_list = nn.ModuleList([self.agg.ConvGRUCell_00])

    # ---- engine lifecycle: parameters live in the engine's flat arena on the current device
    def _ensure_engine(self, block: torch.Tensor):
        B = block.shape[0]
        dev = block.device
        key = (B, dev, self.compute_dtype)
        first = self.backbone.conv1.weight
        if self._engine is not None and self._engine_key == key and first.data_ptr() == self._engine.PRM[self._param_names[0]].data_ptr():
            return
        named = {k: v for k, v in self.named_parameters()}
        bad = [k for k in self._param_names if named[k].dtype != torch.float32]
        if bad:   # .half() / .double() / .to(torch.bfloat16) cast the Parameters themselves; the arena holds f32 master weights
            raise TypeError(f"dpc_amd.DPC_RNN keeps float32 master parameters (the reference's state_dict dtype); {bad[0]} is "
                            f"{named[bad[0]].dtype}.  Select the kernels' operand type with model.bfloat16() / model.float() (or "
                            "DPC_COMPUTE_DTYPE=bf16|f32); .half() and .double() are not supported")
        eng = DPCEngine(self.network, self.sample_size, self.num_seq, self.seq_len, self.pred_step, B, dev,
                        self.compute_dtype, self.widths, lib=self._simulator)
        eng.load_params({k: named[k].detach() for k in self._param_names})
        if self._engine is not None:
            # a rebuild (another batch size, model.bfloat16() after training started, a device move) continues the SAME optimisation:
            # Adam moments, step counter / bias corrections and the dropout draw counter move over, as they would stay put under
            # torch.optim.Adam, whose state is keyed on the Parameter objects and survives all three
            eng.adopt_optimizer_state(self._engine)
        for k in self._param_names:
            named[k].data = eng.PRM[k]  # re-point the Parameter at its slice of the flat arena
        self._engine, self._engine_key = eng, key
        self.mask = None

    def forward(self, block):
        # block: [B, N, C, SL, H, W] (dpc/model_3d.py:47-49)
        if getattr(self, "_is_replica", False):
            return self._replica_forward(block)
        if block.device.type != "cuda" and self._simulator is None:
            raise L.DpcError("dpc_amd.DPC_RNN runs on MI355X only: move the module and the input to a cuda (HIP) device")
        self._ensure_engine(block)
        self._engine.packed_for_step = -1  # parameters may have been changed by an external optimizer
        named = dict(self.named_parameters())
        params = [named[k] for k in self._param_names]
        score = _DPCScore.apply(self, block.float(), self.training, self._forced_masks, *params)
        if self.mask is None:  # only compute mask once (model_3d.py:86-96); contiguous
            self.mask = self._engine.get_mask()
        return [score, self.mask]

    def _replica_forward(self, block):
        """forward of an nn.DataParallel replica (see _REPLICA_NOTE): per-device engine from the root's registry, parameters = the
        broadcast copies torch hung on this replica, gradients back through them"""
        if block.device.type != "cuda" and self._simulator is None:
            raise L.DpcError("dpc_amd.DPC_RNN runs on MI355X only: move the module and the input to a cuda (HIP) device")
        dev, B = block.device, block.shape[0]
        slot = getattr(self, "_replica_slot", dev.index or 0)   # tests on the simulator tell two "devices" apart by slot
        params = []
        for k in self._param_names:   # the broadcast copies are plain attributes of the replicated holder modules (replicate.py)
            obj = self
            for part in k.split("."):
                obj = getattr(obj, part)
            params.append(obj)
        key = (str(dev), slot, B, self.compute_dtype)
        with self._replica_lock:
            eng = self._replica_engines.get(key)
            if eng is None:
                if not self._replica_engines:
                    warnings.warn(_REPLICA_NOTE, stacklevel=3)
                eng = DPCEngine(self.network, self.sample_size, self.num_seq, self.seq_len, self.pred_step, B, dev, self.compute_dtype,
                                self.widths, lib=self._simulator, seed=233 + slot)   # per-replica dropout stream (dpc/model_3d.py:18 seeds 233)
                for old_key in [q for q in self._replica_engines if q[:2] == key[:2]]:   # another batch size / dtype on this device: one engine per device
                    del self._replica_engines[old_key]
                self._replica_engines[key] = eng
        masks = self._forced_masks
        if isinstance(masks, (list, tuple)):
            masks = masks[slot]
        score = _DPCScoreReplica.apply(eng, tuple(self._param_names), block.float(), self.training, masks, *params)
        return [score, eng.get_mask()]

    def reset_mask(self):
        self.mask = None

    # ---- compute dtype through the calls a user of the reference would make.  Parameters stay f32 (master weights of the
    # fused Adam, the reference's state_dict dtype); only the kernels' operand type changes, and the engine is rebuilt at the
    # next forward.  model.bfloat16() = throughput mode (BASELINE configs[1]), model.float() = the 1e-3 parity mode.
    def bfloat16(self):
        self.compute_dtype = torch.bfloat16
        return self

    def float(self):
        self.compute_dtype = torch.float32
        return self

    def half(self):
        raise TypeError("dpc_amd.DPC_RNN: fp16 is not a mode of this build (bf16 operands with f32 accumulation and f32 master "
                        "weights are: model.bfloat16())")

    def double(self):
        raise TypeError("dpc_amd.DPC_RNN: parameters are float32 master weights; there is no f64 mode")

    def to(self, *args, **kwargs):
        """device moves as nn.Module.to; a floating dtype selects the compute dtype like .bfloat16() / .float() instead of casting
        the f32 master parameters (model.to(torch.bfloat16) == model.bfloat16())"""
        dtype = kwargs.get("dtype")          # only an EXPLICIT dtype argument is a compute-dtype selection; everything else
        rest = [a for a in args if not isinstance(a, torch.dtype)]   # (device, tensor, memory_format, non_blocking) goes to nn.Module.to
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
        if dtype is not None:
            if dtype not in (torch.float32, torch.bfloat16):
                raise TypeError(f"dpc_amd.DPC_RNN.to({dtype}): compute dtypes are torch.float32 and torch.bfloat16")
            self.compute_dtype = dtype
        kw = {k: v for k, v in kwargs.items() if k != "dtype"}
        if rest and isinstance(rest[0], torch.Tensor):   # model.to(tensor): its device only, never its dtype
            rest[0] = rest[0].device
        return super().to(*rest, **kw) if (rest or kw) else self

    # ---- copies.  The engine holds device buffers, ctypes descriptors and captured graphs: none of that is copied or pickled.
    def __deepcopy__(self, memo):
        """a fresh module with the same constructor arguments, the same parameter VALUES (clones, not arena views) and no engine --
        it builds its own at its first forward (copy.deepcopy(model): EMA / averaged copies, torch.optim.swa_utils.AveragedModel)"""
        new = type(self)(self.sample_size, self.num_seq, self.seq_len, self.pred_step, self.network, compute_dtype=self.compute_dtype,
                         widths=self.widths, _simulator=self._simulator)
        new.compute_dtype = self.compute_dtype   # the constructor lets DPC_COMPUTE_DTYPE override its argument; a copy keeps what the original RUNS in
        mine = dict(self.named_parameters())
        with torch.no_grad():
            for k, q in new.named_parameters():
                q.data = mine[k].detach().clone()
                q.requires_grad_(mine[k].requires_grad)
        new.train(self.training)
        memo[id(self)] = new
        return new

    def __reduce_ex__(self, protocol):
        raise TypeError("dpc_amd.DPC_RNN is not picklable as a whole (its engine owns device buffers and captured graphs): save "
                        "model.state_dict() as dpc/main.py:166-174 does (dpc_amd.checkpoint writes the reference's dictionary)")

    def _score_buffers(self, eng):
        if self._score_bufs is None or self._score_bufs[0].shape != eng.score.shape or self._score_bufs[0].device != eng.score.device:
            self._score_bufs = (eng.score, torch.empty_like(eng.score))
        return self._score_bufs

    @property
    def engine(self) -> Optional[DPCEngine]:
        return self._engine
