"""Downstream action classifier ``LC`` (eval/model_3d_lc.py:12-65, driven by eval/test.py) on the MI355X kernels --
SURVEY.md section 8 f3.

Same 2d3d-ResNet and ConvGRU kernels as the DPC-RNN step; what differs from ``DPC_RNN``:
  * the backbone's BatchNorm3d layers track running statistics (``track_running_stats=True``, model_3d_lc.py:27-29):
    batch statistics + buffer update in train mode, running buffers in eval mode;
  * ReLU BEFORE the temporal mean (model_3d_lc.py:52-54), then the ConvGRU aggregates ALL num_seq blocks and only its last
    state is used (``context[:, -1]``, model_3d_lc.py:57-58);
  * head: spatial mean -> BatchNorm1d -> Dropout(p) -> Linear(num_class); loss = CrossEntropyLoss on the clip label,
    accuracy = top-1 (eval/test.py:244-255).

``LCEngine`` is the static-schedule engine (forward / backward / fused Adam on flat arenas, like ``DPCEngine``);
``LC`` is the drop-in ``nn.Module`` with the reference's constructor, ``forward(block) -> (output, context)`` and
state_dict keys (running buffers included), so ``--pretrain`` of a DPC checkpoint and ``--test`` files load by key.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .engine import BN_EPS, DPCEngine, LAYER_PLAN, LAYER_WIDTH, param_shapes


def lc_param_shapes(network: str, num_class: int, widths: Sequence[int] = LAYER_WIDTH) -> "Dict[str, Tuple[int, ...]]":
    """trainable parameters of LC in the reference's registration order (model_3d_lc.py:29-45)"""
    base = param_shapes(network, widths)
    out = {k: v for k, v in base.items() if not k.startswith("network_pred.")}
    D = widths[3]
    out["final_bn.weight"] = (D,)
    out["final_bn.bias"] = (D,)
    out["final_fc.1.weight"] = (num_class, D)
    out["final_fc.1.bias"] = (num_class,)
    return out


def lc_state_dict_keys(network: str, num_class: int, widths: Sequence[int] = LAYER_WIDTH):
    """every state_dict key of the reference LC, in order: parameters interleaved with the BatchNorm buffers
    (weight, bias, running_mean, running_var, num_batches_tracked), the ConvGRU alias keys after the cell's own"""
    keys = []
    shapes = lc_param_shapes(network, num_class, widths)
    names = list(shapes)
    for i, k in enumerate(names):
        keys.append(k)
        is_bn_bias = k.endswith(".bias") and (".bn" in k or k.startswith("backbone.bn1") or "downsample.1" in k or k.startswith("final_bn"))
        if is_bn_bias:
            pre = k[: -len("bias")]
            keys += [pre + "running_mean", pre + "running_var", pre + "num_batches_tracked"]
        if k == "agg.ConvGRUCell_00.out_gate.bias":
            keys += [n.replace("agg.ConvGRUCell_00.", "agg.cell_list.0.") for n in names if n.startswith("agg.ConvGRUCell_00.")]
    return keys


class LCEngine(DPCEngine):
    """forward / loss / backward / Adam of the LC classifier as fixed kernel sequences (no autograd graph)"""
    BN_RUNNING = True

    def __init__(self, network: str = "resnet18", sample_size: int = 128, num_seq: int = 8, seq_len: int = 5, batch: int = 4,
                 device="cuda", compute_dtype=torch.float32, widths: Sequence[int] = LAYER_WIDTH, lib: Optional[L.Lib] = None,
                 lr: float = 1e-3, wd: float = 1e-3, dropout: float = 0.5, num_class: int = 101, gru_dropout: float = 0.1,
                 seed: int = 666):
        self.num_class = int(num_class)
        self.p_drop_fc = float(dropout)
        super().__init__(network, sample_size, num_seq, seq_len, 0, batch, device, compute_dtype, widths, lib, lr, wd, gru_dropout,
                         seed, score_path="materialised")

    def _param_shapes(self):
        return lc_param_shapes(self.network, self.num_class, self.widths)

    # ---- head: ConvGRU over all N blocks (P = 0) + BatchNorm1d / Dropout / Linear / CE
    def _build_head(self):
        B, N, SQ, D, dt, NC = self.B, self.N, self.SQ, self.D, self.cdtype, self.num_class
        f32 = torch.float32
        if D % 32 or D > 256:
            raise ValueError("feature size must be a multiple of 32, at most 256 (fused ConvGRU recurrence)")
        M = B * SQ
        self.M, self.n_agg, self.n_steps = M, N, N
        self.feat_relu = self.empty((N, M, D), dt)
        self.X_all = self.feat_relu
        self.H_all = torch.zeros((N + 1, M, D), dtype=dt, device=self.device)
        self.HR_all = self.empty((N, M, D), dt)
        self.G_all = self.empty((N, M, 3 * D), dt)
        self.U_all, self.R_all, self.O_all = (self.empty((N, M, D), f32) for _ in range(3))
        self.d_featrelu = self.empty((N, M, D), f32)
        self.d_feat = self.empty(self.feat_shape + (D,), dt)
        self.d_hlast = self.empty((M, D), f32)
        self.gru_packed = self.empty((16 * D * D,), dt)
        self.gru_ws = self.empty((2, M, D), f32)
        self.dev_step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.dev_bc = torch.ones(2, dtype=f32, device=self.device)
        self.dev_draw = torch.zeros(1, dtype=torch.int32, device=self.device)  # dropout draw counter (one per train-mode forward)
        self.drop_all = None
        self.dWx, self.dWh, self.dWo = self.empty((3 * D, D), f32), self.empty((2 * D, D), f32), self.empty((D, D), f32)
        self.db = self.empty((3 * D,), f32)
        for (co, kk) in ((3 * D, D), (2 * D, D), (D, D)):
            self._need_wgrad(N * M, co, kk)
        self.need_part(64 * 3 * D)
        self.BUF["final_bn.running_mean"] = torch.zeros(D, dtype=f32, device=self.device)
        self.BUF["final_bn.running_var"] = torch.ones(D, dtype=f32, device=self.device)
        self.BUF["final_bn.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=self.device)
        self.target = torch.zeros(B, dtype=torch.int64, device=self.device)
        self.ctx, self.xhat, self.context, self.y_drop, self.dctx = (self.empty((B, D), f32) for _ in range(5))
        self.bn1d_stat = self.empty((2, D), f32)
        self.logits, self.dlogits = self.empty((B, NC), f32), self.empty((B, NC), f32)
        self.row_ws, self.result = self.empty((B, 2), f32), self.empty((2,), f32)
        Pm = self.PRM
        gd = self.gru_desc = L.GruChainDesc()
        gd.dtype, gd.M, gd.D, gd.SQ, gd.P, gd.n_agg, gd.n_steps = L.dtype_code(dt), M, D, SQ, 0, N, N
        gd.p_drop, gd.seed = float(self.p_drop), self.seed
        for name, t in (("packed", self.gru_packed), ("bias_u", Pm["agg.ConvGRUCell_00.update_gate.bias"]),
                        ("bias_r", Pm["agg.ConvGRUCell_00.reset_gate.bias"]), ("bias_o", Pm["agg.ConvGRUCell_00.out_gate.bias"]),
                        ("bias_1", Pm["agg.ConvGRUCell_00.out_gate.bias"]), ("bias_2", Pm["agg.ConvGRUCell_00.out_gate.bias"]),  # unused (P = 0)
                        ("X_all", self.X_all), ("H_all", self.H_all), ("HR_all", self.HR_all), ("U_all", self.U_all),
                        ("R_all", self.R_all), ("O_all", self.O_all), ("G_all", self.G_all), ("d_x", self.d_featrelu),
                        ("ws", self.gru_ws), ("d_hlast", self.d_hlast)):
            setattr(gd, name, t.data_ptr())
        hd = self.head_desc = L.LcHeadDesc()
        hd.dtype, hd.B, hd.SQ, hd.D, hd.num_class, hd.train = L.dtype_code(dt), B, SQ, D, NC, 1
        hd.p_drop, hd.momentum, hd.eps, hd.seed = self.p_drop_fc, self.BN_MOMENTUM, BN_EPS, self.seed  # own Philox stream id, same key
        G = self.G
        for name, t in (("h_last", self.H_all[N]), ("bn_weight", Pm["final_bn.weight"]), ("bn_bias", Pm["final_bn.bias"]),
                        ("bn_running_mean", self.BUF["final_bn.running_mean"]), ("bn_running_var", self.BUF["final_bn.running_var"]),
                        ("bn_num_batches", self.BUF["final_bn.num_batches_tracked"]), ("fc_weight", Pm["final_fc.1.weight"]),
                        ("fc_bias", Pm["final_fc.1.bias"]), ("target", self.target), ("ctx", self.ctx), ("xhat", self.xhat),
                        ("bn_out", self.context), ("y", self.y_drop), ("stat", self.bn1d_stat), ("logits", self.logits),
                        ("dlogits", self.dlogits), ("row_ws", self.row_ws), ("result", self.result),
                        ("g_fc_weight", G["final_fc.1.weight"]), ("g_fc_bias", G["final_fc.1.bias"]),
                        ("g_bn_weight", G["final_bn.weight"]), ("g_bn_bias", G["final_bn.bias"]), ("dctx", self.dctx),
                        ("d_hlast", self.d_hlast)):
            setattr(hd, name, t.data_ptr())

    # ---- state_dict with the running buffers, reference key order
    def load_params(self, params: Dict[str, torch.Tensor]):
        own = {k: v for k, v in params.items() if (k[7:] if k.startswith("module.") else k) not in self.BUF}
        super().load_params(own)
        for k, v in params.items():
            k2 = k[7:] if k.startswith("module.") else k
            if k2 in self.BUF:
                self.BUF[k2].copy_(v.to(self.BUF[k2].dtype).reshape(self.BUF[k2].shape))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {}
        for k in lc_state_dict_keys(self.network, self.num_class, self.widths):
            if k in self.PRM:
                out[k] = self.PRM[k].detach().clone()
            elif k in self.BUF:
                out[k] = self.BUF[k].detach().clone()
            else:
                out[k] = out[k.replace("agg.cell_list.0.", "agg.ConvGRUCell_00.")]
        return out

    # ---- forward (model_3d_lc.py:47-65) + loss / accuracy (eval/test.py:244-255)
    def forward(self, block: torch.Tensor, target: torch.Tensor, train: bool = False, gru_masks: Optional[torch.Tensor] = None,
                fc_mask: Optional[torch.Tensor] = None):
        """block [B,N,3,SL,H,W] f32, target [B] int64 class labels.  Returns (output [B,1,num_class], context [B,1,D]) as the
        reference does; ``self.result`` = device f32[2] (mean CE loss, top-1 accuracy).  gru_masks [N,M,D] / fc_mask [B,D]:
        optional explicit pre-scaled dropout masks (tests); train=True without them draws Philox masks in the kernels."""
        B, N, SQ, D, M = self.B, self.N, self.SQ, self.D, self.M
        if block is not None and (tuple(block.shape) != (B, N, 3, self.SL, self.size, self.size) or block.dtype != torch.float32):
            raise ValueError(f"block must be float32 [B,N,3,SL,H,W] = {(B, N, 3, self.SL, self.size, self.size)}")
        self.train_mode = bool(train)
        self.target.copy_(target.reshape(B).to(self.device, torch.int64))
        x = self._backbone_forward(block)
        dc = L.dtype_code(self.cdtype)
        self.call("dpc_relu_tpool_fwd", x, dc, B, N, self.feat_shape[1], SQ, D, self.feat_relu)
        gd, hd = self.gru_desc, self.head_desc
        self._keep = (gru_masks, fc_mask)  # explicit masks must outlive the launches
        if gru_masks is not None:
            gru_masks = gru_masks.to(self.device, torch.float32).contiguous()
            self._keep = (gru_masks, fc_mask)
        gd.drop_masks = gru_masks.data_ptr() if gru_masks is not None else None
        if train and (gru_masks is None or fc_mask is None):
            self.call("dpc_counter_advance", self.dev_draw)
        gd.step_dev = self.dev_draw.data_ptr() if (gru_masks is None and train and self.p_drop > 0) else None
        self.call("dpc_gru_chain_fwd", C.byref(gd))
        if fc_mask is not None:
            fc_mask = fc_mask.to(self.device, torch.float32).contiguous()
            self._keep = (gru_masks, fc_mask)
        hd.train = int(train)
        hd.drop_mask = fc_mask.data_ptr() if fc_mask is not None else None
        hd.step_dev = self.dev_draw.data_ptr()
        self.call("dpc_lc_head_fwd", C.byref(hd))
        return self.logits.view(B, 1, self.num_class), self.context.view(B, 1, D)

    def backward(self, on_tail_ready=None):
        """gradients of the mean CE loss of the last train-mode forward w.r.t. every parameter"""
        if not self.train_mode:
            raise L.DpcError("LCEngine.backward needs a train-mode forward (BatchNorm with batch statistics)")
        B, N, SQ, D, M = self.B, self.N, self.SQ, self.D, self.M
        dc = L.dtype_code(self.cdtype)
        self.call("dpc_lc_head_bwd", C.byref(self.head_desc))
        self.call("dpc_gru_chain_bwd", C.byref(self.gru_desc))
        Gm = self.G
        self.gemm_tn(self.G_all, 3 * D, self.X_all, D, self.dWx, N * M, 3 * D, D)
        self.gemm_tn(self.G_all, 3 * D, self.H_all, D, self.dWh, N * M, 2 * D, D)
        self.gemm_tn(self.G_all[:, :, 2 * D:], 3 * D, self.HR_all, D, self.dWo, N * M, D, D)
        self.call("dpc_colsum", self.G_all, dc, 3 * D, N * M, 3 * D, self.db, 0, self.part, self.part.numel())
        for i, (g, n) in enumerate((("u", "update_gate"), ("r", "reset_gate"), ("o", "out_gate"))):
            w = Gm[f"agg.ConvGRUCell_00.{n}.weight"].view(D, 2 * D)
            self.call("dpc_copy2d_f32", self.dWx[i * D:(i + 1) * D], D, w, 2 * D, D, D)
            self.call("dpc_copy2d_f32", self.dWh[i * D:(i + 1) * D] if g != "o" else self.dWo, D, w[:, D:], 2 * D, D, D)
            self.call("dpc_copy2d_f32", self.db[i * D:(i + 1) * D], D, Gm[f"agg.ConvGRUCell_00.{n}.bias"], D, 1, D)
        self.call("dpc_relu_tpool_bwd", self.blocks[-1].out, self.d_featrelu, dc, B, N, self.feat_shape[1], SQ, D, self.d_feat)
        self._backbone_backward(self.d_feat, on_tail_ready)

    def train_step(self, block: torch.Tensor, target: torch.Tensor, allreduce=None, **masks) -> torch.Tensor:
        """forward + CE / accuracy + backward (+ gradient all-reduce) + Adam; returns device f32[2] = loss, top-1"""
        self.forward(block, target, train=True, **masks)
        if allreduce is not None and hasattr(allreduce, "start"):
            self.backward(on_tail_ready=allreduce.start)
            allreduce.finish(self.flat_g[:self.grad_split])
        else:
            self.backward()
            if allreduce is not None:
                allreduce(self.flat_g)
        self.adam_step()
        return self.result

    # the DPC-specific entry points do not exist on this engine
    def loss_topk(self, with_grad: bool = True):
        return self.result

    def capture_train_step(self, *a, **k):
        raise NotImplementedError("hipGraph capture is wired for the DPC-RNN step only")

    def get_mask(self):
        raise AttributeError("LC has no contrastive mask")


class _Holder(nn.Module):
    """namespace module so that parameters / buffers appear under the reference's dotted names"""


def _attach(root: nn.Module, dotted: str, value, buffer: bool):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            setattr(mod, p, _Holder())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], value)
    else:
        mod.register_parameter(parts[-1], value)


class LC(nn.Module):
    """drop-in for eval/model_3d_lc.py:12: ``LC(sample_size, num_seq, seq_len, network, dropout, num_class)``,
    ``forward(block) -> (output [B,1,num_class], context [B,1,D])``.  Inference and the engine's own train step are the
    supported uses (``engine.train_step(block, target)``); torch autograd through forward is not wired for this head."""

    def __init__(self, sample_size, num_seq, seq_len, network="resnet18", dropout=0.5, num_class=101,
                 compute_dtype=torch.float32, widths=LAYER_WIDTH, seed: int = 0, _simulator: Optional[L.Lib] = None):
        super().__init__()
        if network not in LAYER_PLAN:
            raise IOError("model type is wrong")
        self.sample_size, self.num_seq, self.seq_len, self.num_class = sample_size, num_seq, seq_len, num_class
        self.network, self.dropout, self.compute_dtype, self.widths = network, dropout, compute_dtype, tuple(widths)
        self.last_duration = int(math.ceil(seq_len / 4))
        self.last_size = int(math.ceil(sample_size / 32))
        self.param = {"feature_size": widths[3], "num_layers": 1, "hidden_size": widths[3]}
        self._simulator, self._engine, self._engine_key = _simulator, None, None
        g = torch.Generator().manual_seed(seed)
        shapes = lc_param_shapes(network, num_class, widths)
        for k in lc_state_dict_keys(network, num_class, widths):
            if k.startswith("agg.cell_list.0."):
                continue
            if k in shapes:
                shp = shapes[k]
                if k.startswith("backbone") and len(shp) == 5:   # kaiming_normal(fan_out), resnet_2d3d.py:224-230
                    v = torch.randn(shp, generator=g) * math.sqrt(2.0 / (shp[0] * shp[2] * shp[3] * shp[4]))
                elif k.startswith("backbone") or k.startswith("final_bn"):
                    v = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
                elif k.endswith("bias"):
                    v = torch.zeros(shp)
                else:                                              # orthogonal gain 1 (model_3d_lc.py:67-72)
                    v = torch.empty(shp)
                    nn.init.orthogonal_(v, 1, generator=g)
                _attach(self, k, nn.Parameter(v), buffer=False)
            else:
                C_ = shapes[k.rsplit(".", 1)[0] + ".weight"][0]
                v = torch.zeros(C_) if k.endswith("running_mean") else (torch.ones(C_) if k.endswith("running_var") else torch.zeros((), dtype=torch.int64))
                _attach(self, k, v, buffer=True)
        self.agg.cell_list = nn.ModuleList([self.agg.ConvGRUCell_00])

    def _ensure_engine(self, block):
        """Parameters and BatchNorm buffers of the module ARE the engine's arenas (as in DPC_RNN): ``load_state_dict`` / an
        optimizer writing through them reach the kernels, and what ``engine.train_step`` / a train-mode forward update (weights,
        running statistics, num_batches_tracked) is what ``state_dict()`` returns."""
        key = (block.shape[0], block.device, self.compute_dtype)
        first = self.backbone.conv1.weight
        if self._engine is not None and self._engine_key == key and first.data_ptr() == self._engine.PRM["backbone.conv1.weight"].data_ptr():
            return
        named = dict(self.named_parameters())
        bad = [k for k, v in named.items() if v.dtype != torch.float32]
        if bad:   # .half() / .double() cast the Parameters themselves; the arena holds f32 master weights (as dpc_amd.DPC_RNN)
            raise TypeError(f"dpc_amd.LC keeps float32 master parameters; {bad[0]} is {named[bad[0]].dtype}.  Select the kernels' operand "
                            "type with model.bfloat16() / model.float(); .half() and .double() are not supported")
        eng = LCEngine(self.network, self.sample_size, self.num_seq, self.seq_len, block.shape[0], block.device, self.compute_dtype,
                       self.widths, self._simulator, dropout=self.dropout, num_class=self.num_class)
        eng.load_params({k: v.detach() for k, v in self.state_dict().items()})
        if self._engine is not None:
            # a rebuild (another batch size, model.bfloat16() after training started, a device move) continues the SAME optimisation:
            # Adam moments, step counter, bias corrections, dropout draw counter, lr / wd (DPC_RNN._ensure_engine does the same)
            eng.adopt_optimizer_state(self._engine)
            eng.train_mode = self._engine.train_mode
        for k, t in eng.PRM.items():
            named[k].data = t  # re-point the Parameter at its slice of the flat arena
        for k, t in eng.BUF.items():
            mod = self
            *path, leaf = k.split(".")
            for p_ in path:
                mod = getattr(mod, p_)
            mod._buffers[leaf] = t  # same tensor object the kernels update
        self._engine, self._engine_key = eng, key

    @property
    def engine(self) -> Optional[LCEngine]:
        return self._engine

    # ---- compute dtype through the calls a user of the reference would make (as dpc_amd.DPC_RNN): parameters stay f32
    def bfloat16(self):
        self.compute_dtype = torch.bfloat16
        return self

    def float(self):
        self.compute_dtype = torch.float32
        return self

    def half(self):
        raise TypeError("dpc_amd.LC: fp16 is not a mode of this build (bf16 operands with f32 accumulation and f32 master weights are: model.bfloat16())")

    def double(self):
        raise TypeError("dpc_amd.LC: parameters are float32 master weights; there is no f64 mode")

    def to(self, *args, **kwargs):
        """device moves as nn.Module.to; an explicit floating dtype selects the compute dtype instead of casting the master parameters"""
        dtype = kwargs.get("dtype")
        rest = [a for a in args if not isinstance(a, torch.dtype)]
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
        if dtype is not None:
            if dtype not in (torch.float32, torch.bfloat16):
                raise TypeError(f"dpc_amd.LC.to({dtype}): compute dtypes are torch.float32 and torch.bfloat16")
            self.compute_dtype = dtype
        kw = {k: v for k, v in kwargs.items() if k != "dtype"}
        return super().to(*rest, **kw) if (rest or kw) else self

    def __deepcopy__(self, memo):
        """a fresh module with the same constructor arguments, cloned parameter / buffer values and no engine (as DPC_RNN.__deepcopy__)"""
        new = type(self)(self.sample_size, self.num_seq, self.seq_len, self.network, self.dropout, self.num_class, self.compute_dtype,
                         self.widths, _simulator=self._simulator)
        new.load_state_dict({k: v.detach().clone() for k, v in self.state_dict().items()})
        mine = dict(self.named_parameters())
        for k, q in new.named_parameters():   # --train_what ft / last freeze parameters: the copy keeps that
            q.requires_grad_(mine[k].requires_grad)
        new.train(self.training)
        memo[id(self)] = new
        return new

    def __reduce_ex__(self, protocol):
        raise TypeError("dpc_amd.LC is not picklable as a whole (its engine owns device buffers): save model.state_dict() as "
                        "eval/test.py:205-214 does")

    def forward(self, block, target: Optional[torch.Tensor] = None):
        if getattr(self, "_is_replica", False):   # nn.DataParallel over several devices (eval/test.py:63): see dpc_amd.model._REPLICA_ERROR
            from .model import _REPLICA_ERROR
            raise RuntimeError(_REPLICA_ERROR.replace("dpc_amd.DPC_RNN", "dpc_amd.LC").replace("dpc_amd.main", "dpc_amd.lc_main").replace("dpc/main.py", "eval/test.py"))
        if block.device.type != "cuda" and self._simulator is None:
            raise L.DpcError("dpc_amd.LC runs on MI355X only: move the module and the input to a cuda (HIP) device")
        self._ensure_engine(block)
        self._engine.packed_for_step = -1  # the parameters may have been changed through the module since the last forward
        if target is None:
            target = torch.zeros(block.shape[0], dtype=torch.int64)
        out, ctx = self._engine.forward(block.float(), target, train=self.training)
        return out.clone(), ctx.clone()
