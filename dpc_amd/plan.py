"""Which gfx950 kernel serves which convolution of the 2d3d-ResNet -- the static schedule's kernel plan, queryable
without a GPU (``dpc_conv_plan`` runs the library's own dispatch code with the launch skipped, include/dpc_hip.h).

``backbone_units`` walks the reference's layer plan (backbone/resnet_2d3d.py:205-284) exactly as ``DPCEngine`` does and
``unit_descs`` builds the three descriptors of a Conv3d + BatchNorm3d unit (forward, input-gradient, weight-gradient): the
engine uses the same function, so the table printed here is what a train step launches.  tests/test_plan.py pins the table
for BASELINE.json's configurations; ``python -m dpc_amd.plan cfg2`` prints it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch

from . import _lib as L

LAYER_PLAN = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}  # backbone/resnet_2d3d.py:274-284
LAYER_WIDTH = (64, 128, 256, 256)                                    # resnet_2d3d.py:217-223


def out_shape_of(in_shape, k, s, p, stem=False):
    N, T, H, W = in_shape
    if stem:
        return (N, T, H, W)  # space-to-depth grid == output grid
    return (N, (T + 2 * p[0] - k[0]) // s[0] + 1, (H + 2 * p[1] - k[1]) // s[1] + 1, (W + 2 * p[2] - k[2]) // s[2] + 1)


def unit_descs(Ci: int, Co: int, k, s, p, in_shape, dtype: torch.dtype, stem: bool = False):
    """(forward, input-gradient, weight-gradient) descriptors of one conv unit on a channels-last [N,T,H,W,Ci] input"""
    N, T, H, W = in_shape
    No, To, Ho, Wo = out_shape_of(in_shape, k, s, p, stem)
    taps = k[0] * k[1] * k[2]
    dc = L.dtype_code(dtype)
    Kp = taps * Ci
    desc_f = L.ConvDesc(dc, dc, 0, N, To, Ho, Wo, T, H, W, Ci, Ci, Co, Kp, Co, *k, *s, *p)
    # input-gradient: rows enumerate the forward input grid, source is dy on the output grid
    desc_d = L.ConvDesc(dc, dc, 1, N, T, H, W, To, Ho, Wo, Co, Co, Ci, taps * Co, Ci, *k, *s, *p)
    desc_w = L.ConvDesc(dc, L.F32, 0, N, To, Ho, Wo, T, H, W, Ci, Ci, Co, Kp, Co, *k, *s, *p)
    return desc_f, desc_d, desc_w


def backbone_units(network: str, size: int, batch: int, num_seq: int = 8, seq_len: int = 5,
                   widths: Sequence[int] = LAYER_WIDTH) -> "List[Dict]":
    """every conv unit of the backbone in forward order: name, geometry and how the backward uses its input-gradient"""
    if network not in LAYER_PLAN:
        raise IOError("model type is wrong")
    BN, H = batch * num_seq, size
    units = [dict(name="conv1", Ci=16, Co=widths[0], k=(1, 4, 4), s=(1, 1, 1), p=(0, 2, 2), in_shape=(BN, seq_len, H // 2, H // 2),
                  stem=True, dgrad=None)]
    hp = (H // 2 - 1) // 2 + 1
    shape, inplanes = (BN, seq_len, hp, hp), widths[0]
    for li in range(4):
        planes = widths[li]
        is3d = li >= 2
        for bi in range(LAYER_PLAN[network][li]):
            stride = 2 if (li > 0 and bi == 0) else 1
            has_ds = bi == 0 and (stride != 1 or inplanes != planes)
            k = (3, 3, 3) if is3d else (1, 3, 3)
            p = (1, 1, 1) if is3d else (0, 1, 1)
            s = (stride,) * 3 if is3d else (1, stride, stride)
            cin = inplanes if bi == 0 else planes
            pre = f"layer{li + 1}.{bi}."
            mid = out_shape_of(shape, k, s, p)
            # c1's input-gradient carries the residual branch as its addend unless a downsample accumulates in place afterwards
            units.append(dict(name=pre + "conv1", Ci=cin, Co=planes, k=k, s=s, p=p, in_shape=shape, stem=False,
                              dgrad="plain" if has_ds else "addend"))
            units.append(dict(name=pre + "conv2", Ci=planes, Co=planes, k=k, s=(1, 1, 1), p=p, in_shape=mid, stem=False, dgrad="plain"))
            if has_ds:
                units.append(dict(name=pre + "downsample.0", Ci=cin, Co=planes, k=(1, 1, 1), s=s, p=(0, 0, 0), in_shape=shape,
                                  stem=False, dgrad="inplace"))
            shape, inplanes = mid, planes
    return units


def ex_supported(lib: L.Lib, desc_d, addend: bool, gate: bool, bnred: bool) -> bool:
    """dpc_conv_igemm_ex serves this input-gradient with the given fused pieces WITHOUT demoting it from a specialised kernel
    to the generic implicit GEMM (the query runs the library's own dispatch code)"""
    ex = L.conv_plan(lib, desc_d, L.PLAN_IGEMM, addend=addend, addend_mask=gate, bnred=bnred)
    plain = L.conv_plan(lib, desc_d, L.PLAN_IGEMM, addend=addend)
    return bool(ex) and (not ex.startswith("igemm_kernel") or plain.startswith("igemm_kernel"))


def fold_table(lib: L.Lib, network: str, size: int, batch: int, dtype: torch.dtype, **kw) -> "List[Tuple[str, bool, bool, bool]]":
    """per BasicBlock: (block, fold_c1, gate, fold_prev) as DPCEngine decides them (engine._Block.plan_backward):
    conv2's input-gradient carries bn1's backward reduction / conv1's input-gradient gates the residual gradient on the fly /
    ... and carries the previous block's bn2 reduction"""
    units = backbone_units(network, size, batch, **kw)
    nblocks = sum(LAYER_PLAN[network])
    rows = []
    last = f"layer4.{LAYER_PLAN[network][3] - 1}."
    first = True
    for u in units:
        if not u["name"].endswith("conv1") or u["name"] == "conv1":
            continue
        pre = u["name"][:-len("conv1")]
        c2 = next(v for v in units if v["name"] == pre + "conv2")
        has_ds = any(v["name"] == pre + "downsample.0" for v in units)
        final_relu = pre != last
        d1 = unit_descs(u["Ci"], u["Co"], u["k"], u["s"], u["p"], u["in_shape"], dtype)[1]
        d2 = unit_descs(c2["Ci"], c2["Co"], c2["k"], c2["s"], c2["p"], c2["in_shape"], dtype)[1]
        fold_c1 = ex_supported(lib, d2, False, False, True)
        gate = (not has_ds) and ex_supported(lib, d1, True, final_relu, False)
        fold_prev = (not first) and gate and ex_supported(lib, d1, True, final_relu, True)
        rows.append((pre[:-1], fold_c1, gate, fold_prev))
        first = False
    assert len(rows) == nblocks
    return rows


def plan_table(lib: L.Lib, network: str, size: int, batch: int, dtype: torch.dtype, **kw) -> "List[Tuple[str, str, str]]":
    """(unit, op, kernel) for every conv launch of one train step's backbone"""
    rows = []
    for u in backbone_units(network, size, batch, **kw):
        f, d, w = unit_descs(u["Ci"], u["Co"], u["k"], u["s"], u["p"], u["in_shape"], dtype, u["stem"])
        rows.append((u["name"], "fwd", L.conv_plan(lib, f, L.PLAN_IGEMM, stats=True)))
        if u["dgrad"] is not None:
            rows.append((u["name"], "dgrad" + ("" if u["dgrad"] == "plain" else "+" + u["dgrad"]),
                         L.conv_plan(lib, d, L.PLAN_IGEMM, addend=u["dgrad"] != "plain")))
        rows.append((u["name"], "wgrad", L.conv_plan(lib, w, L.PLAN_WGRAD, dy_ld=u["Co"])))
    return rows


CONFIGS = {"cfg2": ("resnet18", 128, 128), "cfg4": ("resnet34", 224, 44), "cfg5": ("resnet34", 224, 64)}


if __name__ == "__main__":
    import sys
    lib = L.Lib(L.HIP_LIB_PATH, "probe")  # dlopen only: the query launches nothing and needs no GPU
    for cfg in sys.argv[1:] or ["cfg2"]:
        net, size, batch = CONFIGS[cfg]
        for dt in (torch.bfloat16, torch.float32):
            print(f"# {cfg}: {net} {size}^2 batch {batch} {dt}")
            for name, op, kern in plan_table(lib, net, size, batch, dt):
                print(f"{name:24s} {op:14s} {kern}")
            for name, a, b, c in fold_table(lib, net, size, batch, dt):
                print(f"{name:24s} fused epilogues: bn1 reduction in conv2's input-gradient {int(a)}, gated residual {int(b)}, "
                      f"previous bn2 reduction {int(c)}")
