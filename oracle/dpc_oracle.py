"""CPU oracle for the DPC-RNN training step.  TEST INFRASTRUCTURE ONLY.

This file is a functional (module-free) torch-CPU fp32/fp64 restatement of the
reference algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``dpc_amd``) never does and fails loudly when its HIP library is missing.

Parity pinning: the reference ships no tests or golden vectors,
so this oracle is pinned against outputs of the reference itself, imported on
CPU in the build container by ``tests/golden/make_golden.py`` (fixtures under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks them).

Every function cites the reference lines (relative to /root/reference) whose
arithmetic it restates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

BN_EPS = 1e-5  # torch.nn.BatchNorm3d default, backbone/resnet_2d3d.py:55,93,212

# backbone/resnet_2d3d.py:274-284 -- BasicBlock plans (2d,2d,3d,3d)
LAYER_PLAN = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}
LAYER_WIDTH = (64, 128, 256, 256)  # backbone/resnet_2d3d.py:217-222 (layer4 stays 256)
FEATURE_SIZE = 256  # backbone/select_backbone.py:7,10


# --------------------------------------------------------------------------
# configuration-derived sizes (dpc/model_3d.py:24-25)
# --------------------------------------------------------------------------
def derived_sizes(sample_size: int, seq_len: int = 5) -> Tuple[int, int]:
    last_duration = int(math.ceil(seq_len / 4))
    last_size = int(math.ceil(sample_size / 32))
    return last_duration, last_size


# --------------------------------------------------------------------------
# parameter inventory: names and shapes of the reference state_dict
# (dpc/model_3d.py:28-40, backbone/resnet_2d3d.py:205-257, backbone/convrnn.py:13-15,55-58)
# --------------------------------------------------------------------------
def param_shapes(network: str = "resnet18", widths: Sequence[int] = LAYER_WIDTH,
                 feature: Optional[int] = None, with_alias: bool = True) -> "Dict[str, Tuple[int, ...]]":
    """Ordered {state_dict key: shape}.  ``widths``/``feature`` default to the
    reference's; narrower values give the width-parametric test nets (the
    reference block classes are width-parametric, resnet_2d3d.py:50,86)."""
    plan = LAYER_PLAN[network]
    feature = widths[3] if feature is None else feature
    out: Dict[str, Tuple[int, ...]] = {}
    out["backbone.conv1.weight"] = (widths[0], 3, 1, 7, 7)
    out["backbone.bn1.weight"] = (widths[0],)
    out["backbone.bn1.bias"] = (widths[0],)
    inplanes = widths[0]
    for li in range(4):
        planes = widths[li]
        is3d = li >= 2
        k = (3, 3, 3) if is3d else (1, 3, 3)
        stride = 1 if li == 0 else 2
        for bi in range(plan[li]):
            pre = f"backbone.layer{li + 1}.{bi}."
            cin = inplanes if bi == 0 else planes
            out[pre + "conv1.weight"] = (planes, cin) + k
            out[pre + "bn1.weight"] = (planes,)
            out[pre + "bn1.bias"] = (planes,)
            out[pre + "conv2.weight"] = (planes, planes) + k
            out[pre + "bn2.weight"] = (planes,)
            out[pre + "bn2.bias"] = (planes,)
            if bi == 0 and (stride != 1 or inplanes != planes):
                out[pre + "downsample.0.weight"] = (planes, inplanes, 1, 1, 1)
                out[pre + "downsample.1.weight"] = (planes,)
                out[pre + "downsample.1.bias"] = (planes,)
        inplanes = planes
    cells = ["agg.ConvGRUCell_00."] + (["agg.cell_list.0."] if with_alias else [])
    for pre in cells:
        for gate in ("reset_gate", "update_gate", "out_gate"):
            out[pre + gate + ".weight"] = (feature, 2 * feature, 1, 1)
            out[pre + gate + ".bias"] = (feature,)
    for idx in (0, 2):
        out[f"network_pred.{idx}.weight"] = (feature, feature, 1, 1)
        out[f"network_pred.{idx}.bias"] = (feature,)
    return out


def make_params_pcg(network: str = "resnet18", widths: Sequence[int] = LAYER_WIDTH,
                    dtype=torch.float32) -> Params:
    """Portable weight recipe used on both sides of every parity test
    (SURVEY.md §8c G3): key i -> PCG64(1000+i) normals; convs scaled
    1/sqrt(fan_in); BN gamma 1+0.1n, beta 0.1n; biases 0.05n; the ConvGRU
    alias keys (backbone/convrnn.py:55-58) share the first cell's tensors."""
    shapes = param_shapes(network, widths)
    params: Params = {}
    for i, (k, shp) in enumerate(shapes.items()):
        if k.startswith("agg.cell_list.0."):
            params[k] = params[k.replace("agg.cell_list.0.", "agg.ConvGRUCell_00.")]
            continue
        n = np.random.Generator(np.random.PCG64(1000 + i)).standard_normal(shp, dtype=np.float32)
        if len(shp) >= 4:
            fan_in = int(np.prod(shp[1:]))
            v = n / np.float32(math.sqrt(fan_in))
        elif ".bn" in k or "downsample.1" in k:
            v = (1.0 + 0.1 * n) if k.endswith("weight") else 0.1 * n
        else:
            v = 0.05 * n
        params[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dtype)
    return params


def make_input_pcg(B: int, num_seq: int, seq_len: int, size: int, seed: int = 7) -> torch.Tensor:
    x = np.random.Generator(np.random.PCG64(seed)).standard_normal(
        (B, num_seq, 3, seq_len, size, size), dtype=np.float32)
    return torch.from_numpy(x)


# --------------------------------------------------------------------------
# backbone (backbone/resnet_2d3d.py)
# --------------------------------------------------------------------------
def bn_batch(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = BN_EPS) -> torch.Tensor:
    """BatchNorm3d with track_running_stats=False (dpc/model_3d.py:28): batch
    statistics (biased variance) in train AND eval (torch functional/batch_norm)."""
    dims = (0, 2, 3, 4)
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    shp = (1, -1, 1, 1, 1)
    return (x - mean) / torch.sqrt(var + eps) * gamma.view(shp) + beta.view(shp)


def basic_block(x: torch.Tensor, p: Params, pre: str, is3d: bool, stride: int,
                final_relu: bool = True) -> torch.Tensor:
    """BasicBlock2d (resnet_2d3d.py:83-116) / BasicBlock3d (:47-80)."""
    if is3d:
        s1, pad = (stride,) * 3, (1, 1, 1)
    else:
        s1, pad = (1, stride, stride), (0, 1, 1)
    out = F.conv3d(x, p[pre + "conv1.weight"], None, s1, pad)
    out = F.relu(bn_batch(out, p[pre + "bn1.weight"], p[pre + "bn1.bias"]))
    out = F.conv3d(out, p[pre + "conv2.weight"], None, 1, pad)
    out = bn_batch(out, p[pre + "bn2.weight"], p[pre + "bn2.bias"])
    if (pre + "downsample.0.weight") in p:  # resnet_2d3d.py:232-245
        res = F.conv3d(x, p[pre + "downsample.0.weight"], None, s1, 0)
        res = bn_batch(res, p[pre + "downsample.1.weight"], p[pre + "downsample.1.bias"])
    else:
        res = x
    out = out + res
    return F.relu(out) if final_relu else out


def ste_round_bf16(x: torch.Tensor) -> torch.Tensor:
    """round to bf16 in the forward, identity in the backward: how a tensor the engine STORES in bf16 enters the next op while the
    gradient flows through unchanged (the engine's backward kernels differentiate the unrounded op at the stored values)"""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def basic_block_rounded(x: torch.Tensor, p: Params, pre: str, is3d: bool, stride: int, final_relu: bool = True) -> torch.Tensor:
    """basic_block with the engine's throughput-mode storage points: weights, raw conv outputs and activations pass through
    ste_round_bf16; BatchNorm statistics are those of the STORED raw outputs, accumulation stays full precision.  The block-local
    gradient checks of the bf16 mode (tests/test_block_grads_gpu.py) differentiate this, not the fp32 block."""
    r = ste_round_bf16
    if is3d:
        s1, pad = (stride,) * 3, (1, 1, 1)
    else:
        s1, pad = (1, stride, stride), (0, 1, 1)
    raw1 = r(F.conv3d(x, r(p[pre + "conv1.weight"]), None, s1, pad))
    act1 = r(F.relu(bn_batch(raw1, p[pre + "bn1.weight"], p[pre + "bn1.bias"])))
    raw2 = r(F.conv3d(act1, r(p[pre + "conv2.weight"]), None, 1, pad))
    out = bn_batch(raw2, p[pre + "bn2.weight"], p[pre + "bn2.bias"])
    if (pre + "downsample.0.weight") in p:
        rawd = r(F.conv3d(x, r(p[pre + "downsample.0.weight"]), None, s1, 0))
        res = bn_batch(rawd, p[pre + "downsample.1.weight"], p[pre + "downsample.1.bias"])
    else:
        res = x
    out = out + res
    return r(F.relu(out) if final_relu else out)


def backbone_forward(p: Params, x: torch.Tensor, network: str = "resnet18") -> torch.Tensor:
    """ResNet2d3d_full.forward (resnet_2d3d.py:259-270); x is [BN,3,T,H,W]."""
    plan = LAYER_PLAN[network]
    x = F.conv3d(x, p["backbone.conv1.weight"], None, (1, 2, 2), (0, 3, 3))
    x = F.relu(bn_batch(x, p["backbone.bn1.weight"], p["backbone.bn1.bias"]))
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    for li in range(4):
        for bi in range(plan[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            last = li == 3 and bi == plan[li] - 1  # resnet_2d3d.py:249-252
            x = basic_block(x, p, f"backbone.layer{li + 1}.{bi}.", li >= 2, stride, final_relu=not last)
    return x


# --------------------------------------------------------------------------
# ConvGRU (backbone/convrnn.py)
# --------------------------------------------------------------------------
def convgru_cell(x: torch.Tensor, h: torch.Tensor, p: Params, pre: str = "agg.ConvGRUCell_00.") -> torch.Tensor:
    """ConvGRUCell.forward (convrnn.py:24-34), kernel_size 1."""
    comb = torch.cat([x, h], dim=1)
    upd = torch.sigmoid(F.conv2d(comb, p[pre + "update_gate.weight"], p[pre + "update_gate.bias"]))
    rst = torch.sigmoid(F.conv2d(comb, p[pre + "reset_gate.weight"], p[pre + "reset_gate.bias"]))
    out = torch.tanh(F.conv2d(torch.cat([x, h * rst], dim=1), p[pre + "out_gate.weight"], p[pre + "out_gate.bias"]))
    return h * (1 - upd) + out * upd


# --------------------------------------------------------------------------
# DPC_RNN.forward (dpc/model_3d.py:46-98)
# --------------------------------------------------------------------------
def dpc_forward(p: Params, block: torch.Tensor, network: str = "resnet18", pred_step: int = 3,
                dropout_masks: Optional[List[torch.Tensor]] = None,
                return_intermediates: bool = False):
    """Returns score [B,P,SQ,B,P,SQ].  ``dropout_masks``: None = eval mode
    (dropout off; BN still batch-stat, model_3d.py:28); else a list of
    (num_seq-pred_step)+pred_step pre-scaled masks [B,D,ls,ls] multiplied onto
    the carried hidden state after every GRU step (convrnn.py:78)."""
    B, N, C, SL, H, W = block.shape
    last_duration, last_size = derived_sizes(H, SL)
    D = p["network_pred.0.weight"].shape[0]
    feat = backbone_forward(p, block.reshape(B * N, C, SL, H, W), network)
    feat = F.avg_pool3d(feat, (last_duration, 1, 1), stride=(1, 1, 1))  # model_3d.py:53
    assert feat.shape[2] == 1 and feat.shape[3] == last_size
    feat_inf_all = feat.view(B, N, D, last_size, last_size)
    feature = F.relu(feat).view(B, N, D, last_size, last_size)
    feature_inf = feat_inf_all[:, N - pred_step:].contiguous()  # model_3d.py:58
    k = 0
    h = torch.zeros(B, D, last_size, last_size, dtype=block.dtype)  # convrnn.py:25-27
    for t in range(N - pred_step):  # model_3d.py:62, convrnn.py:76-79
        h = convgru_cell(feature[:, t], h, p)
        if dropout_masks is not None:
            h = h * dropout_masks[k]
        k += 1
    preds = []
    for _ in range(pred_step):  # model_3d.py:66-71
        pt = F.conv2d(h, p["network_pred.0.weight"], p["network_pred.0.bias"])
        pt = F.conv2d(F.relu(pt), p["network_pred.2.weight"], p["network_pred.2.bias"])
        preds.append(pt)
        h = convgru_cell(F.relu(pt), h, p)
        if dropout_masks is not None:
            h = h * dropout_masks[k]
        k += 1
    pred = torch.stack(preds, 1)
    SQ = last_size ** 2
    pred2 = pred.permute(0, 1, 3, 4, 2).reshape(B * pred_step * SQ, D)  # model_3d.py:81
    feat2 = feature_inf.permute(0, 1, 3, 4, 2).reshape(B * pred_step * SQ, D)  # model_3d.py:82
    score = torch.matmul(pred2, feat2.t()).view(B, pred_step, SQ, B, pred_step, SQ)  # model_3d.py:83
    if return_intermediates:
        return score, {"pred": pred2, "feature_inf": feat2, "backbone": feat}
    return score


# --------------------------------------------------------------------------
# mask / target / loss / top-k (dpc/model_3d.py:86-96, dpc/main.py:178-185,213-218,
# utils/utils.py:38-55)
# --------------------------------------------------------------------------
def mask_closed_form(B: int, P: int, SQ: int) -> torch.Tensor:
    """int8 [B,P,SQ,B,P,SQ]: 1 pos (b=b',s=s',p=n); -1 temporal neg (b=b',s=s',p!=n);
    -3 spatial neg (b=b',s!=s'); 0 easy neg."""
    b = torch.arange(B)
    pp = torch.arange(P)
    s = torch.arange(SQ)
    same_b = (b.view(B, 1, 1, 1, 1, 1) == b.view(1, 1, 1, B, 1, 1))
    same_s = (s.view(1, 1, SQ, 1, 1, 1) == s.view(1, 1, 1, 1, 1, SQ))
    same_p = (pp.view(1, P, 1, 1, 1, 1) == pp.view(1, 1, 1, 1, P, 1))
    m = torch.zeros(B, P, SQ, B, P, SQ, dtype=torch.int8)
    m[(same_b & ~same_s).expand_as(m)] = -3
    m[(same_b & same_s & ~same_p).expand_as(m)] = -1
    m[(same_b & same_s & same_p).expand_as(m)] = 1
    return m


def mask_by_reference_steps(B: int, P: int, SQ: int) -> torch.Tensor:
    """Step-by-step restatement of model_3d.py:86-96 (note Q9: N==P there)."""
    N = P
    m = torch.zeros(B, P, SQ, B, N, SQ, dtype=torch.int8)
    for k in range(B):
        m[k, :, :, k, :, :] = -3
    for k in range(B):
        for s in range(SQ):
            m[k, :, s, k, :, s] = -1
    t = m.permute(0, 2, 1, 3, 5, 4).contiguous().view(B * SQ, P, B * SQ, N)
    for j in range(B * SQ):
        for q in range(P):
            t[j, q, j, N - P + q] = 1
    return t.view(B, SQ, P, B, SQ, N).permute(0, 2, 1, 3, 5, 4).contiguous()


def target_from_mask(mask: torch.Tensor) -> torch.Tensor:
    """process_output + argmax (main.py:178-185,213-215)."""
    B, P, SQ, B2, NS, _ = mask.shape
    return (mask == 1).reshape(B * P * SQ, B2 * NS * SQ).to(torch.int64).argmax(dim=1)


def loss_and_topk(score: torch.Tensor, target: Optional[torch.Tensor] = None):
    """CrossEntropyLoss(mean) + calc_topk_accuracy (main.py:217-218, utils.py:38-55)."""
    B, P, SQ = score.shape[:3]
    rows = B * P * SQ
    flat = score.reshape(rows, -1)
    if target is None:
        target = torch.arange(rows)
    loss = F.cross_entropy(flat, target)
    _, pred = flat.topk(5, 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    accs = [correct[:k].reshape(-1).float().sum() / rows for k in (1, 3, 5)]
    return loss, accs


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr: float = 1e-3, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, wd: float = 1e-5):
    """torch.optim.Adam with L2 weight decay folded into the gradient
    (main.py:80-81; torch/optim/adam.py single-tensor path). In place."""
    g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def unique_param_names(p: Params) -> List[str]:
    return [k for k in p if not k.startswith("agg.cell_list.0.")]


def train_step_reference(p: Params, block: torch.Tensor, network: str = "resnet18", pred_step: int = 3,
                         dropout_masks: Optional[List[torch.Tensor]] = None):
    """forward + CE + top-k + backward (main.py:198-231, without the optimizer).
    Returns loss, accs, {name: grad} (alias keys share the cell's grad)."""
    names = unique_param_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    full = dict(leaves)
    for k in p:
        if k.startswith("agg.cell_list.0."):
            full[k] = leaves[k.replace("agg.cell_list.0.", "agg.ConvGRUCell_00.")]
    score = dpc_forward(full, block, network, pred_step, dropout_masks)
    loss, accs = loss_and_topk(score)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), [a.item() for a in accs], dict(zip(names, grads)), score.detach()


def init_params_reference_style(network: str = "resnet18", seed: int = 0,
                                widths: Sequence[int] = LAYER_WIDTH) -> Params:
    """Reference initialisation: kaiming_normal(fan_out) convs, BN 1/0
    (resnet_2d3d.py:224-230); orthogonal gain 1 / zero bias for agg and
    network_pred (model_3d.py:100-106)."""
    g = torch.Generator().manual_seed(seed)
    params: Params = {}
    for k, shp in param_shapes(network, widths).items():
        if k.startswith("agg.cell_list.0."):
            params[k] = params[k.replace("agg.cell_list.0.", "agg.ConvGRUCell_00.")]
        elif k.startswith("backbone") and len(shp) == 5:
            fan_out = shp[0] * int(np.prod(shp[2:]))
            params[k] = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_out)
        elif k.startswith("backbone"):
            params[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        elif k.endswith("bias"):
            params[k] = torch.zeros(shp)
        else:
            w = torch.empty(shp)
            torch.nn.init.orthogonal_(w, 1, generator=g)
            params[k] = w
    return params


# --------------------------------------------------------------------------
# LC downstream classifier (eval/model_3d_lc.py:12-65, eval/test.py:244-255) -- SURVEY.md section 8 f3
# --------------------------------------------------------------------------
BN_MOMENTUM = 0.1  # torch.nn.BatchNorm default


def lc_state_dict_keys(network: str = "resnet18", num_class: int = 101, widths: Sequence[int] = LAYER_WIDTH) -> List[str]:
    """state_dict keys of the reference LC, in its order (BatchNorm buffers interleaved; backbone built with
    track_running_stats=True, model_3d_lc.py:27-29; ConvGRU alias keys, convrnn.py:55-58)"""
    keys: List[str] = []
    base = [k for k in param_shapes(network, widths, with_alias=True) if not k.startswith("network_pred.")]
    for k in base:
        keys.append(k)
        if k.startswith("backbone.") and k.endswith(".bias"):
            pre = k[: -len("bias")]
            keys += [pre + "running_mean", pre + "running_var", pre + "num_batches_tracked"]
    keys += ["final_bn.weight", "final_bn.bias", "final_bn.running_mean", "final_bn.running_var", "final_bn.num_batches_tracked",
             "final_fc.1.weight", "final_fc.1.bias"]
    return keys


def lc_shapes(network: str = "resnet18", num_class: int = 101, widths: Sequence[int] = LAYER_WIDTH) -> "Dict[str, Tuple[int, ...]]":
    base = param_shapes(network, widths, with_alias=True)
    D = widths[3]
    out: Dict[str, Tuple[int, ...]] = {}
    for k in lc_state_dict_keys(network, num_class, widths):
        if k in base:
            out[k] = base[k]
        elif k.endswith("num_batches_tracked"):
            out[k] = ()
        elif k.startswith("final_fc.1."):
            out[k] = (num_class, D) if k.endswith("weight") else (num_class,)
        elif k.startswith("final_bn."):
            out[k] = (D,)
        else:  # running_mean / running_var of a backbone BatchNorm3d: its channel count
            out[k] = base[k.rsplit(".", 1)[0] + ".weight"]
    return out


def make_lc_params_pcg(network: str = "resnet18", num_class: int = 101, widths: Sequence[int] = LAYER_WIDTH) -> Params:
    """portable recipe for every LC state_dict entry: as make_params_pcg (key i -> PCG64(3000+i)); running_mean 0.1 n,
    running_var 1 + 0.1 |n|, num_batches_tracked 0, the Linear scaled 1/sqrt(fan_in)"""
    shapes = lc_shapes(network, num_class, widths)
    params: Params = {}
    for i, (k, shp) in enumerate(shapes.items()):
        if k.startswith("agg.cell_list.0."):
            params[k] = params[k.replace("agg.cell_list.0.", "agg.ConvGRUCell_00.")]
            continue
        if k.endswith("num_batches_tracked"):
            params[k] = torch.zeros((), dtype=torch.int64)
            continue
        n = np.random.Generator(np.random.PCG64(3000 + i)).standard_normal(shp, dtype=np.float32)
        if k.endswith("running_mean"):
            v = 0.1 * n
        elif k.endswith("running_var"):
            v = 1.0 + 0.1 * np.abs(n)
        elif len(shp) >= 2:
            v = n / np.float32(math.sqrt(int(np.prod(shp[1:]))))
        elif ".bn" in k or "downsample.1" in k or k.startswith("final_bn"):
            v = (1.0 + 0.1 * n) if k.endswith("weight") else 0.1 * n
        else:
            v = 0.05 * n
        params[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return params


def _bn_running(x: torch.Tensor, p: Params, new: Params, pre: str, train: bool, dims, shp, eps: float = BN_EPS) -> torch.Tensor:
    """BatchNorm with track_running_stats=True (torch.nn.functional.batch_norm semantics): train = batch statistics and
    running <- (1-m) running + m batch (unbiased variance); eval = running statistics"""
    g, b = p[pre + "weight"].view(shp), p[pre + "bias"].view(shp)
    if train:
        mean = x.mean(dims, keepdim=True)
        var = x.var(dims, unbiased=False, keepdim=True)
        n = x.numel() / x.shape[1]
        with torch.no_grad():
            new[pre + "running_mean"] = (1 - BN_MOMENTUM) * p[pre + "running_mean"] + BN_MOMENTUM * mean.reshape(-1).to(torch.float32)
            new[pre + "running_var"] = (1 - BN_MOMENTUM) * p[pre + "running_var"] + BN_MOMENTUM * (var.reshape(-1) * n / max(n - 1, 1)).to(torch.float32)
            new[pre + "num_batches_tracked"] = p[pre + "num_batches_tracked"] + 1
    else:
        mean, var = p[pre + "running_mean"].view(shp).to(x.dtype), p[pre + "running_var"].view(shp).to(x.dtype)
    return (x - mean) / torch.sqrt(var + eps) * g + b


def _lc_block(x, p, new, pre, is3d, stride, final_relu, train):
    """BasicBlock2d / 3d with running-stat BatchNorm (resnet_2d3d.py:47-116, track_running_stats=True)"""
    s1, pad = ((stride,) * 3, (1, 1, 1)) if is3d else ((1, stride, stride), (0, 1, 1))
    bn = lambda t, name: _bn_running(t, p, new, pre + name + ".", train, (0, 2, 3, 4), (1, -1, 1, 1, 1))
    out = F.relu(bn(F.conv3d(x, p[pre + "conv1.weight"], None, s1, pad), "bn1"))
    out = bn(F.conv3d(out, p[pre + "conv2.weight"], None, 1, pad), "bn2")
    if (pre + "downsample.0.weight") in p:
        res = bn(F.conv3d(x, p[pre + "downsample.0.weight"], None, s1, 0), "downsample.1")
    else:
        res = x
    out = out + res
    return F.relu(out) if final_relu else out


def lc_forward(p: Params, block: torch.Tensor, network: str = "resnet18", train: bool = False,
               gru_masks: Optional[List[torch.Tensor]] = None, fc_mask: Optional[torch.Tensor] = None):
    """LC.forward (model_3d_lc.py:47-65).  Returns output [B,1,num_class], context [B,1,D] and the dict of updated buffers
    (train mode).  gru_masks: pre-scaled masks [B,D,ls,ls] per ConvGRU step (None = no dropout); fc_mask [B,D] likewise."""
    B, N, Cc, SL, H, W = block.shape
    last_duration, last_size = derived_sizes(H, SL)
    D = p["final_bn.weight"].shape[0]
    new: Params = {}
    plan = LAYER_PLAN[network]
    x = block.reshape(B * N, Cc, SL, H, W)
    x = F.conv3d(x, p["backbone.conv1.weight"], None, (1, 2, 2), (0, 3, 3))
    x = F.relu(_bn_running(x, p, new, "backbone.bn1.", train, (0, 2, 3, 4), (1, -1, 1, 1, 1)))
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    for li in range(4):
        for bi in range(plan[li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            last = li == 3 and bi == plan[li] - 1
            x = _lc_block(x, p, new, f"backbone.layer{li + 1}.{bi}.", li >= 2, stride, not last, train)
    feature = F.avg_pool3d(F.relu(x), (last_duration, 1, 1), stride=1)  # model_3d_lc.py:52-54: ReLU first
    feature = feature.view(B, N, D, last_size, last_size)
    h = torch.zeros(B, D, last_size, last_size, dtype=block.dtype)
    for t in range(N):  # convrnn.py:76-79
        h = convgru_cell(feature[:, t], h, p)
        if gru_masks is not None:
            h = h * gru_masks[t]
    context = h.mean(dim=(2, 3))  # model_3d_lc.py:58-60: avg_pool3d over the spatial extent of the LAST state
    context = _bn_running(context.unsqueeze(-1), p, new, "final_bn.", train, (0, 2), (1, -1, 1)).squeeze(-1)  # BatchNorm1d on [B,C,1]
    y = context * fc_mask if fc_mask is not None else context
    output = y @ p["final_fc.1.weight"].t() + p["final_fc.1.bias"]
    return output.view(B, 1, -1), context.view(B, 1, D), new


def lc_loss_acc(output: torch.Tensor, target: torch.Tensor):
    """eval/test.py:244-255: CrossEntropyLoss on [B*N', C] logits with the clip label repeated, top-1 accuracy"""
    B, Np, Cn = output.shape
    flat = output.reshape(B * Np, Cn)
    tgt = target.view(B, 1).repeat(1, Np).view(-1)
    loss = F.cross_entropy(flat, tgt)
    acc = (flat.argmax(1) == tgt).float().mean()
    return loss, acc


def lc_train_step_reference(p: Params, block: torch.Tensor, target: torch.Tensor, network: str = "resnet18",
                            gru_masks=None, fc_mask=None):
    """train-mode forward + loss + backward; returns loss, acc, {name: grad}, output, context, updated buffers"""
    names = [k for k in p if not k.startswith("agg.cell_list.0.") and p[k].dtype.is_floating_point and
             not k.endswith(("running_mean", "running_var"))]
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    full = dict(p)
    full.update(leaves)
    for k in p:
        if k.startswith("agg.cell_list.0."):
            full[k] = leaves[k.replace("agg.cell_list.0.", "agg.ConvGRUCell_00.")]
    output, context, new = lc_forward(full, block, network, True, gru_masks, fc_mask)
    loss, acc = lc_loss_acc(output, target)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), acc.item(), dict(zip(names, grads)), output.detach(), context.detach(), new
