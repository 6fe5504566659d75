"""Checkpoint compatibility with the reference.

The reference saves, once per epoch (dpc/main.py:166-174 through utils/utils.py:14-26),

    {'epoch', 'net', 'state_dict': DataParallel(model).state_dict()   # 'module.'-prefixed, alias keys included
     'best_acc', 'optimizer': torch.optim.Adam(...).state_dict(), 'iteration'}

and reads it back strictly on ``--resume`` (model AND optimizer, dpc/main.py:88-102) or by key
intersection on ``--pretrain`` (backbone/resnet_2d3d.py:310-333).  This module produces and consumes
exactly that dictionary from the engine's flat f32 arenas, so a file written here resumes in the
reference and a file written by the reference resumes here -- Adam moments and step count included.

The Adam state is laid out as ``torch.optim.Adam.state_dict()`` lays it out: ``state[i]`` for the i-th
entry of ``model.parameters()`` (the reference's registration order == the arena order, alias
parameters deduplicated) holds ``step`` (0-dim f32 tensor), ``exp_avg`` and ``exp_avg_sq`` with the
parameter's shape; ``param_groups`` is one group whose hyper-parameter keys are taken from the
installed torch (so ``optimizer.load_state_dict`` of that torch accepts it).
"""
from __future__ import annotations

import glob
import os
import warnings
from typing import Dict, Iterable, List, Tuple

import torch

ALIAS_FROM, ALIAS_TO = "agg.ConvGRUCell_00.", "agg.cell_list.0."  # backbone/convrnn.py:55-58


def _strip(k: str) -> str:
    return k[7:] if k.startswith("module.") else k


def model_state_dict(eng, prefix: str = "module.") -> "Dict[str, torch.Tensor]":
    """reference-layout state_dict of the engine's parameters (CPU copies; alias keys share storage)"""
    out: Dict[str, torch.Tensor] = {}
    for k, v in eng.PRM.items():
        out[prefix + k] = v.detach().cpu().clone()
    # the reference lists the alias keys right after the cell's own keys (ModuleList registered second)
    ordered: Dict[str, torch.Tensor] = {}
    cell = [k for k in out if k.startswith(prefix + ALIAS_FROM)]
    for k, v in out.items():
        ordered[k] = v
        if cell and k == cell[-1]:
            for c in cell:
                ordered[c.replace(ALIAS_FROM, ALIAS_TO)] = out[c]
    return ordered


def _adam_group_defaults(lr: float, wd: float) -> dict:
    """hyper-parameter keys/values torch.optim.Adam(params, lr=lr, weight_decay=wd) of the installed torch writes"""
    probe = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=lr, weight_decay=wd)
    g = dict(probe.state_dict()["param_groups"][0])
    g.pop("params")
    return g


def optimizer_state_dict(eng) -> dict:
    """the engine's fused-Adam state in torch.optim.Adam.state_dict() layout"""
    names = list(eng.PRM.keys())
    state = {}
    if eng.step_count > 0:  # torch creates the per-parameter state lazily at the first step
        for i, k in enumerate(names):
            o, n = eng.offsets[k]
            state[i] = {"step": torch.tensor(float(eng.step_count)),
                        "exp_avg": eng.flat_m[o:o + n].detach().cpu().clone().view(eng.shapes[k]),
                        "exp_avg_sq": eng.flat_v[o:o + n].detach().cpu().clone().view(eng.shapes[k])}
    group = _adam_group_defaults(eng.lr, eng.wd)
    group["params"] = list(range(len(names)))
    return {"state": state, "param_groups": [group]}


def load_optimizer_state(eng, opt_sd: dict, use_saved_lr: bool = True) -> None:
    """inverse of optimizer_state_dict; accepts what the reference's torch.optim.Adam wrote"""
    names = list(eng.PRM.keys())
    groups = opt_sd["param_groups"]
    ids = [i for g in groups for i in g["params"]]
    if len(ids) != len(names):
        raise ValueError(f"loaded state dict contains {len(ids)} parameters, the model has {len(names)} "
                         "(torch.optim.Optimizer.load_state_dict raises the same way)")
    g0 = groups[0]
    if g0.get("amsgrad") or g0.get("maximize"):
        raise ValueError("amsgrad / maximize checkpoints are not supported (the reference uses neither, dpc/main.py:80-81)")
    if tuple(g0.get("betas", (0.9, 0.999))) != (0.9, 0.999) or float(g0.get("eps", 1e-8)) != 1e-8:
        raise ValueError("non-default Adam betas / eps in the checkpoint (the reference uses the defaults)")
    # validate everything BEFORE touching the arenas: a rejected file must leave the engine's Adam state as it was
    steps, todo = set(), []
    for pos, i in enumerate(ids):
        st = opt_sd["state"].get(i)
        if st is None:
            continue
        k = names[pos]
        for field in ("exp_avg", "exp_avg_sq"):
            if tuple(st[field].shape) != tuple(eng.shapes[k]):
                raise ValueError(f"optimizer state {i} ({field}) has shape {tuple(st[field].shape)}, parameter {k} is {eng.shapes[k]}")
        steps.add(int(float(st["step"])))
        todo.append((k, st))
    if len(steps) > 1:
        raise ValueError(f"per-parameter Adam step counts differ ({sorted(steps)}): the fused optimizer keeps one")
    eng.flat_m.zero_()
    eng.flat_v.zero_()
    for k, st in todo:
        o, n = eng.offsets[k]
        eng.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1).to(torch.float32))
        eng.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1).to(torch.float32))
    eng.step_count = steps.pop() if steps else 0
    if use_saved_lr:  # optimizer.load_state_dict restores the group's hyper-parameters (dpc/main.py:97-98)
        eng.lr = float(g0["lr"])
        eng.wd = float(g0.get("weight_decay", eng.wd))


def load_model_state(eng, state_dict: "Dict[str, torch.Tensor]", strict: bool = True) -> Tuple[List[str], List[str]]:
    """strict=True: nn.Module.load_state_dict semantics (dpc/main.py:96) -- raises RuntimeError on missing or
    unexpected keys or a shape mismatch.  strict=False: neq_load_customized (backbone/resnet_2d3d.py:310-333) --
    keys present on both sides are loaded, the rest is reported.  Returns (missing, unexpected)."""
    sd = {_strip(k): v for k, v in state_dict.items()}
    bufs = getattr(eng, "BUF", {})  # BatchNorm running buffers of the LC classifier's engine (empty for DPC-RNN)
    mine = set(eng.PRM.keys()) | set(bufs.keys())
    aliases = {k.replace(ALIAS_FROM, ALIAS_TO) for k in mine if k.startswith(ALIAS_FROM)}
    unexpected = [k for k in sd if k not in mine and k not in aliases]
    missing = [k for k in list(eng.PRM.keys()) + list(bufs.keys()) + sorted(aliases) if k not in sd]
    if strict and (missing or unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict for {'LC' if bufs else 'DPC_RNN'}:\n\tMissing key(s) in state_dict: "
                           f"{missing}.\n\tUnexpected key(s) in state_dict: {unexpected}.")
    load = {}
    for k, v in sd.items():
        tgt = k.replace(ALIAS_TO, ALIAS_FROM) if k in aliases else k
        if tgt not in mine:
            continue
        want = tuple(eng.shapes[tgt]) if tgt in eng.shapes else tuple(bufs[tgt].shape)
        if tuple(v.shape) != want:
            raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(v.shape)} from checkpoint, "
                               f"the shape in current model is {want}.")
        if tgt in load and k in aliases:
            continue  # the cell's own key wins over its alias (they hold the same tensor in reference files)
        load[tgt] = v
    eng.load_params(load)
    return missing, unexpected


def build_state(eng, epoch: int, net: str, best_acc: float, iteration: int) -> dict:
    """the dictionary of dpc/main.py:167-172 (epoch = the NEXT epoch to run, as the reference stores epoch+1)"""
    return {"epoch": epoch, "net": net, "state_dict": model_state_dict(eng), "best_acc": best_acc,
            "optimizer": optimizer_state_dict(eng), "iteration": iteration}


def save_checkpoint(state: dict, is_best: bool = False, gap: int = 1, filename: str = "models/checkpoint.pth.tar",
                    keep_all: bool = False) -> None:
    """utils/utils.py:14-26: write, drop the previous epoch's file, keep ONE model_best_* file"""
    torch.save(state, filename)
    d = os.path.dirname(filename)
    if not keep_all:
        last = os.path.join(d, "epoch%s.pth.tar" % str(state["epoch"] - gap))
        if os.path.exists(last):
            os.remove(last)
    if is_best:
        for old in glob.glob(os.path.join(d, "model_best_*.pth.tar")):
            try:
                os.remove(old)
            except OSError:
                pass
        torch.save(state, os.path.join(d, "model_best_epoch%s.pth.tar" % str(state["epoch"])))


def resume(eng, path: str, reset_lr: bool = False) -> dict:
    """--resume (dpc/main.py:88-102): strict model load, optimizer state unless reset_lr.  Returns the bookkeeping
    fields {'epoch', 'iteration', 'best_acc'}."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    load_model_state(eng, ck["state_dict"], strict=True)
    if not reset_lr:
        if "optimizer" in ck:
            load_optimizer_state(eng, ck["optimizer"])
        elif "optimizer_flat" in ck:  # files written by round 1 of this build
            st = ck["optimizer_flat"]
            eng.flat_m.copy_(st["m"])
            eng.flat_v.copy_(st["v"])
            eng.step_count = int(st["step"])
        else:
            warnings.warn(f"{path} holds no optimizer state: Adam moments and step count start from zero")
    return {"epoch": ck["epoch"], "iteration": ck.get("iteration", 0), "best_acc": ck.get("best_acc", 0.0)}


def pretrain(eng, path: str, log=print) -> dict:
    """--pretrain (dpc/main.py:104-112): key-intersection load, with the reference's report of unused keys"""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    missing, unexpected = load_model_state(eng, ck["state_dict"], strict=False)
    log("\n=======Check Weights Loading======")
    log("Weights not used from pretrained file:")
    for k in unexpected:
        log(k)
    log("---------------------------")
    log("Weights not loaded into new model:")
    for k in missing:
        log(k)
    log("===================================\n")
    return {"epoch": ck.get("epoch", 0)}
