"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The fixtures are data: inputs are regenerated on both sides from
the PCG64 recipe in oracle/dpc_oracle.py, expected outputs are what the
reference's own code (dpc/model_3d.py, backbone/*.py, utils/utils.py) produced
on CPU.  Shims applied here and nowhere else:
  * torch.Tensor.cuda -> identity   (hard-coded .cuda() at model_3d.py:88, convrnn.py:27)
  * sys.path += reference backbone/, dpc/, utils/ ; empty `torchvision.transforms` stub
    so utils/utils.py imports (only calc_topk_accuracy is used).

usage:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

torch.Tensor.cuda = lambda self, *a, **k: self
sys.path += [f"{REF}/backbone", f"{REF}/dpc", f"{REF}/utils"]
tv = types.ModuleType("torchvision")
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules.setdefault("torchvision", tv)
sys.modules.setdefault("torchvision.transforms", tv.transforms)

from model_3d import DPC_RNN  # noqa: E402
from convrnn import ConvGRUCell  # noqa: E402
from resnet_2d3d import BasicBlock2d, BasicBlock3d  # noqa: E402
import utils as ref_utils  # noqa: E402

from oracle import dpc_oracle as O  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def build_ref(network, size, pred_step=3):
    m = DPC_RNN(sample_size=size, num_seq=8, seq_len=5, pred_step=pred_step, network=network)
    sd = O.make_params_pcg(network)
    missing = m.load_state_dict(sd, strict=True)
    print("load_state_dict:", missing)
    return m, sd


class InjectedDropout(nn.Module):
    """Stands in for agg.dropout_layer (convrnn.py:59,78): multiplies by a
    recorded, pre-scaled Bernoulli(0.9)/0.9 mask so the run is reproducible."""

    def __init__(self, seed, p=0.1):
        super().__init__()
        self.g = torch.Generator().manual_seed(seed)
        self.p = p
        self.masks = []

    def forward(self, x):
        keep = (torch.rand(x.shape, generator=self.g) >= self.p).to(x.dtype)
        self.masks.append(keep.clone())
        return x * keep / (1 - self.p)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()},
          os.path.getsize(path + ("" if path.endswith(".npz") else ".npz")) // 1024, "KiB")


# ---------------------------------------------------------------- G1/G2 mask + target
def golden_mask():
    out = {}
    for (B, P, SQ) in [(4, 3, 16), (3, 5, 49), (2, 1, 4)]:
        size = int(np.sqrt(SQ)) * 32
        m = DPC_RNN(sample_size=size, num_seq=8, seq_len=5, pred_step=P, network="resnet18")

        class Stub(nn.Module):  # mask depends only on (B,P,SQ): skip the conv stack
            def forward(self, x, ls=int(np.sqrt(SQ))):
                return torch.zeros(x.shape[0], 256, 2, ls, ls)
        m.backbone = Stub()
        m.eval()
        with torch.no_grad():
            _, mask = m(torch.zeros(B, 8, 3, 5, size, size))
        mask = mask.contiguous()
        tgt = (mask == 1).view(B * P * SQ, -1).to(int).argmax(dim=1)  # main.py:183,213-215
        key = f"{B}_{P}_{SQ}"
        out["sha_" + key] = np.frombuffer(hashlib.sha256(mask.numpy().tobytes()).digest(), dtype=np.uint8)
        vals, counts = np.unique(mask.numpy(), return_counts=True)
        out["vals_" + key] = vals
        out["counts_" + key] = counts
        out["target_" + key] = tgt.numpy().astype(np.int64)
        if (B, P, SQ) == (2, 1, 4):
            out["mask_" + key] = mask.numpy()
    save("mask.npz", **out)


# ---------------------------------------------------------------- G3 eval-mode scores
def golden_eval_scores():
    out = {}
    for tag, net, size, B in [("r18_64_b2", "resnet18", 64, 2), ("r18_128_b4", "resnet18", 128, 4),
                              ("r34_64_b2", "resnet34", 64, 2)]:
        m, _ = build_ref(net, size)
        m.eval()
        x = O.make_input_pcg(B, 8, 5, size)
        with torch.no_grad():
            score, _ = m(x)
        out["score_" + tag] = score.numpy().astype(np.float32)
        print(tag, "score absmax", float(score.abs().max()), "std", float(score.std()))
        if tag == "r18_64_b2":
            m64 = m.double()
            with torch.no_grad():
                s64, _ = m64(x.double())
            out["noise_fp32_vs_fp64_" + tag] = np.array(float((s64 - score.double()).abs().max()))
            flat = score.view(B * 3 * 4, -1)
            tgt = torch.arange(flat.shape[0])
            loss = nn.CrossEntropyLoss()(flat, tgt)
            t1, t3, t5 = ref_utils.calc_topk_accuracy(flat, tgt, (1, 3, 5))
            out["evalloss_" + tag] = np.array([loss.item(), t1.item(), t3.item(), t5.item()])
    save("eval_scores.npz", **out)


# ---------------------------------------------------------------- G5/G6 train-mode
def golden_train():
    out = {}
    net, size, B = "resnet18", 64, 2
    x = O.make_input_pcg(B, 8, 5, size)
    # (a) dropout p=0: forward + CE + topk + backward + one Adam step
    m, sd = build_ref(net, size)
    m.train()
    m.agg.dropout_layer.p = 0.0
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)  # main.py:80-81
    score, mask = m(x)
    flat = score.view(B * 3 * 4, -1)
    tgt = (mask.contiguous() == 1).view(flat.shape).to(int).argmax(dim=1)
    loss = nn.CrossEntropyLoss()(flat, tgt)
    t1, t3, t5 = ref_utils.calc_topk_accuracy(flat, tgt, (1, 3, 5))
    opt.zero_grad()
    loss.backward()
    out["score_p0"] = score.detach().numpy()
    out["loss_topk_p0"] = np.array([loss.item(), t1.item(), t3.item(), t5.item()])
    names = [k for k, _ in m.named_parameters()]
    out["param_names"] = np.array(names)
    out["grad_norm_p0"] = np.array([p.grad.norm().item() for _, p in m.named_parameters()])
    out["grad_head_p0"] = np.stack([p.grad.flatten()[:8].numpy() if p.numel() >= 8 else np.zeros(8, np.float32)
                                    for _, p in m.named_parameters()])
    for k in ["backbone.conv1.weight", "backbone.layer3.0.conv1.weight", "backbone.layer4.1.bn2.weight",
              "agg.ConvGRUCell_00.out_gate.weight", "network_pred.2.bias"]:
        gfull = dict(m.named_parameters())[k].grad.flatten()
        stride = max(1, gfull.numel() // 4096)  # strided subsample keeps the fixture small
        out["grad_sub_p0::" + k] = gfull[::stride].numpy().copy()
        out["grad_substride_p0::" + k] = np.array(stride)
    opt.step()
    out["adam_sum_p0"] = np.array([p.detach().double().sum().item() for _, p in m.named_parameters()])
    out["adam_head_p0"] = np.stack([p.detach().flatten()[:8].numpy() if p.numel() >= 8 else np.zeros(8, np.float32)
                                    for _, p in m.named_parameters()])
    # (b) injected dropout masks (keep-bits), forward only
    m, _ = build_ref(net, size)
    m.train()
    inj = InjectedDropout(seed=1234)
    m.agg.dropout_layer = inj
    with torch.no_grad():
        score, _ = m(x)
    keep = torch.stack(inj.masks).numpy().astype(np.uint8)  # [8,B,256,2,2]
    out["drop_keep_bits"] = np.packbits(keep.reshape(-1))
    out["drop_keep_shape"] = np.array(keep.shape)
    out["score_drop"] = score.numpy()
    save("train.npz", **out)


def golden_eval_p5():
    """cfg5's combination (resnet34 + pred_step 5) at a CPU-sized image/batch, eval mode (own file: the round-1
    fixtures stay byte-identical)."""
    out = {}
    m, _ = build_ref("resnet34", 64, 5)
    m.eval()
    x = O.make_input_pcg(2, 8, 5, 64)
    with torch.no_grad():
        score, mask = m(x)
    out["score_r34_64_b2_p5"] = score.numpy().astype(np.float32)
    flat = score.view(2 * 5 * 4, -1)
    tgt = (mask.contiguous() == 1).view(flat.shape).to(int).argmax(dim=1)
    loss = nn.CrossEntropyLoss()(flat, tgt)
    t1, t3, t5 = ref_utils.calc_topk_accuracy(flat, tgt, (1, 3, 5))
    out["evalloss_r34_64_b2_p5"] = np.array([loss.item(), t1.item(), t3.item(), t5.item()])
    save("eval_scores_p5.npz", **out)


# ---------------------------------------------------------------- bf16 anchor
def golden_anchor16():
    golden_anchor("resnet18", 128, 16, "anchor_r18_128_b16.npz")


def golden_anchor34():
    """the 224^2 family (BASELINE configs[3], [4]: resnet34, 56^2 / 28^2 / 14x14x3 / 7x7x2 planes) -- round 6"""
    golden_anchor("resnet34", 224, 4, "anchor_r34_224_b4.npz", sub=512)


def golden_anchor(net, size, B, fname, sub=1024):
    """fp32 reference run at a batch at which the engine's throughput (bf16) mode selects
    every specialised kernel -- so the bf16 path is anchored to the reference instead of to itself."""
    out = {}
    x = O.make_input_pcg(B, 8, 5, size)
    m, _ = build_ref(net, size)
    m.train()
    m.agg.dropout_layer.p = 0.0
    score, mask = m(x)
    R = B * 3 * (-(-size // 32)) ** 2
    flat = score.view(R, -1)
    tgt = (mask.contiguous() == 1).view(flat.shape).to(int).argmax(dim=1)
    loss = nn.CrossEntropyLoss()(flat, tgt)
    t1, t3, t5 = ref_utils.calc_topk_accuracy(flat, tgt, (1, 3, 5))
    loss.backward()
    out["score_stride"] = np.array(37)
    out["score_sub"] = score.detach().flatten()[::37].numpy().copy()
    out["score_norm"] = np.array(score.detach().norm().item())
    out["score_diag"] = flat.detach().diagonal().numpy().copy()
    out["loss_topk"] = np.array([loss.item(), t1.item(), t3.item(), t5.item()])
    names = [k for k, _ in m.named_parameters()]
    out["param_names"] = np.array(names)
    out["grad_norm"] = np.array([p.grad.norm().item() for _, p in m.named_parameters()])
    for k, p in m.named_parameters():
        g = p.grad.flatten()
        stride = max(1, g.numel() // sub)
        out["grad_sub::" + k] = g[::stride].numpy().copy()
        out["grad_substride::" + k] = np.array(stride)
    # ---- how far bf16 storage alone moves these outputs: the SAME reference model with every Conv3d / BatchNorm3d /
    # ReLU / MaxPool3d / Conv2d output (and the gradient flowing back through it) rounded to bf16 by hooks.  This is
    # the noise level a correct bf16 implementation shows against the fp32 run; the GPU test's tolerances are
    # multiples of it (per parameter), not guesses.
    class RoundBF16(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.bfloat16().float()

        @staticmethod
        def backward(ctx, g):
            return g.bfloat16().float()

    m2, _ = build_ref(net, size)
    m2.train()
    m2.agg.dropout_layer.p = 0.0
    for mod in m2.modules():
        if isinstance(mod, (nn.Conv3d, nn.BatchNorm3d, nn.ReLU, nn.MaxPool3d, nn.Conv2d)):
            mod.register_forward_hook(lambda mod_, inp, o: RoundBF16.apply(o))
    score2, _ = m2(x.bfloat16().float())
    loss2 = nn.CrossEntropyLoss()(score2.view(R, -1), tgt)
    loss2.backward()
    out["noise_score_l2"] = np.array(((score2 - score).norm() / score.norm()).item())
    out["noise_loss"] = np.array(abs(loss2.item() - loss.item()))
    ref_g = dict(m.named_parameters())
    out["noise_grad_l2"] = np.array([((p2.grad - ref_g[k].grad).norm() / ref_g[k].grad.norm()).item() for k, p2 in m2.named_parameters()])
    out["noise_grad_norm"] = np.array([abs(p2.grad.norm().item() / ref_g[k].grad.norm().item() - 1.0) for k, p2 in m2.named_parameters()])
    print("bf16-rounding noise of the reference: score", float(out["noise_score_l2"]), "grad rel-L2 max", float(out["noise_grad_l2"].max()),
          "grad-norm max", float(out["noise_grad_norm"].max()))
    save(fname, **out)


# ---------------------------------------------------------------- checkpoint layout
def golden_ckpt_layout():
    """Structure (key names / order / shapes / dtypes, optimizer layout) of the dictionary the reference saves at
    dpc/main.py:166-174, built with the reference's own model class wrapped as dpc/main.py:65,80-81 wrap it.
    58 MB of weights are not a fixture; the layout is."""
    import json
    out = {}
    for net in ("resnet18", "resnet34"):
        m = DPC_RNN(sample_size=128, num_seq=8, seq_len=5, pred_step=3, network=net)
        model = nn.DataParallel(m)                                                   # main.py:65
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)      # main.py:80-81
        for p_ in model.parameters():
            p_.grad = torch.zeros_like(p_)
        opt.step()
        state = {"epoch": 1, "net": net, "state_dict": model.state_dict(), "best_acc": 0.0,
                 "optimizer": opt.state_dict(), "iteration": 0}                      # main.py:167-172
        osd = state["optimizer"]
        out[net] = {
            "top_keys": list(state.keys()),
            "state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in state["state_dict"].items()],
            "optimizer_param_order": [k for k, _ in model.named_parameters()],
            "optimizer_state_keys": sorted(osd["state"][0].keys()),
            "optimizer_state_shapes": [[list(osd["state"][i]["exp_avg"].shape), list(osd["state"][i]["step"].shape),
                                        str(osd["state"][i]["step"].dtype)] for i in range(len(osd["state"]))],
            "param_groups": [{k: (v if k != "params" else len(v)) for k, v in g_.items()} for g_ in osd["param_groups"]],
        }
    with open(os.path.join(HERE, "ckpt_layout.json"), "w") as f:
        json.dump(out, f, indent=0, default=lambda o: list(o) if isinstance(o, tuple) else str(o))
    print("wrote ckpt_layout.json", os.path.getsize(os.path.join(HERE, "ckpt_layout.json")) // 1024, "KiB")


# ---------------------------------------------------------------- LC downstream classifier
def golden_lc():
    """eval/model_3d_lc.py LC (reference class, imported): state_dict layout, eval-mode output / context, and a train-mode
    step with both dropouts disabled: loss, accuracy, gradient norms + subsamples, updated running statistics."""
    import json
    sys.path.append(f"{REF}/eval")
    from model_3d_lc import LC
    out = {}
    m = LC(sample_size=64, num_seq=8, seq_len=5, network="resnet18", dropout=0.5, num_class=101)
    sd = O.make_lc_params_pcg("resnet18", 101)
    print("LC load_state_dict:", m.load_state_dict(sd, strict=True))
    layout = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "lc_layout.json"), "w") as f:
        json.dump({"resnet18_101": layout, "param_order": [k for k, _ in m.named_parameters()]}, f, indent=0)
    x = O.make_input_pcg(2, 8, 5, 64)
    target = torch.tensor([[3], [77]])
    m.eval()
    with torch.no_grad():
        o, c = m(x)
    out["eval_output"], out["eval_context"] = o.numpy(), c.numpy()
    m.train()
    m.agg.dropout_layer.p = 0.0
    m.final_fc[0].p = 0.0
    o, c = m(x)
    B, N_, Dn = o.size()
    flat = o.view(B * N_, Dn)
    tgt = target.repeat(1, N_).view(-1)                    # eval/test.py:250-251
    loss = nn.CrossEntropyLoss()(flat, tgt)
    acc = (flat.argmax(1) == tgt).float().mean()
    loss.backward()
    out["train_output"], out["train_context"] = o.detach().numpy(), c.detach().numpy()
    out["train_loss_acc"] = np.array([loss.item(), acc.item()])
    names = [k for k, _ in m.named_parameters()]
    out["param_names"] = np.array(names)
    out["grad_norm"] = np.array([p_.grad.norm().item() for _, p_ in m.named_parameters()])
    for k, p_ in m.named_parameters():
        g_ = p_.grad.flatten()
        st = max(1, g_.numel() // 512)
        out["grad_sub::" + k] = g_[::st].numpy().copy()
        out["grad_substride::" + k] = np.array(st)
    new_sd = m.state_dict()
    for k in ("backbone.bn1", "backbone.layer3.0.bn1", "backbone.layer4.1.bn2", "final_bn"):
        out["rm::" + k] = new_sd[k + ".running_mean"].numpy().copy()
        out["rv::" + k] = new_sd[k + ".running_var"].numpy().copy()
        out["nbt::" + k] = np.array(int(new_sd[k + ".num_batches_tracked"]))
    save("lc.npz", **out)


# ---------------------------------------------------------------- G4 per-op fixtures
def golden_ops():
    out = {}
    g = torch.Generator().manual_seed(42)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    def block_case(tag, cls, cin, cout, stride, shape, with_ds, final_relu=True):
        ds = None
        if with_ds:
            cs = (1, stride, stride) if cls is BasicBlock2d else stride
            ds = nn.Sequential(nn.Conv3d(cin, cout, kernel_size=1, stride=cs, bias=False),
                               nn.BatchNorm3d(cout, track_running_stats=False))
        blk = cls(cin, cout, stride, ds, track_running_stats=False, use_final_relu=final_relu)
        for k, v in blk.state_dict().items():
            v.copy_(rnd(*v.shape) * (0.2 if v.dim() > 1 else 0.5) + (1.0 if (v.dim() == 1 and k.endswith("weight")) else 0.0))
        x = rnd(*shape).requires_grad_(True)
        y = blk(x)
        gy = rnd(*y.shape)
        y.backward(gy)
        out[f"{tag}::x"] = x.detach().numpy()
        out[f"{tag}::y"] = y.detach().numpy()
        out[f"{tag}::gy"] = gy.numpy()
        out[f"{tag}::gx"] = x.grad.numpy()
        for k, v in blk.named_parameters():
            out[f"{tag}::w::{k}"] = v.detach().numpy()
            out[f"{tag}::gw::{k}"] = v.grad.numpy()

    block_case("bb2d_s2_ds", BasicBlock2d, 8, 16, 2, (2, 8, 3, 8, 8), True)
    block_case("bb2d_s1", BasicBlock2d, 8, 8, 1, (2, 8, 2, 6, 6), False)
    block_case("bb3d_s2_ds", BasicBlock3d, 8, 16, 2, (2, 8, 5, 8, 8), True)
    block_case("bb3d_s1_norelu", BasicBlock3d, 8, 8, 1, (2, 8, 3, 4, 4), False, final_relu=False)

    cell = ConvGRUCell(8, 8, 1)
    for k, v in cell.state_dict().items():
        v.copy_(rnd(*v.shape) * 0.3)
    x = rnd(3, 8, 3, 3).requires_grad_(True)
    h = rnd(3, 8, 3, 3).requires_grad_(True)
    hn = cell(x, h)
    gh = rnd(*hn.shape)
    hn.backward(gh)
    out["gru::x"], out["gru::h"], out["gru::hn"] = x.detach().numpy(), h.detach().numpy(), hn.detach().numpy()
    out["gru::gh"], out["gru::gx"], out["gru::ghprev"] = gh.numpy(), x.grad.numpy(), h.grad.numpy()
    for k, v in cell.named_parameters():
        out[f"gru::w::{k}"] = v.detach().numpy()
        out[f"gru::gw::{k}"] = v.grad.numpy()

    sc = rnd(24, 24)
    sc[torch.arange(0, 24, 3), torch.arange(0, 24, 3)] += 3.0
    tgt = torch.arange(24)
    accs = ref_utils.calc_topk_accuracy(sc, tgt, (1, 3, 5))
    out["topk::score"] = sc.numpy()
    out["topk::acc"] = np.array([a.item() for a in accs])
    out["topk::loss"] = np.array(nn.CrossEntropyLoss()(sc, tgt).item())
    save("ops.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["mask", "ops", "eval", "eval_p5", "train", "anchor", "anchor34", "ckpt", "lc"]
    if "mask" in which:
        golden_mask()
    if "ops" in which:
        golden_ops()
    if "eval" in which:
        golden_eval_scores()
    if "train" in which:
        golden_train()
    if "eval_p5" in which:
        golden_eval_p5()
    if "ckpt" in which:
        golden_ckpt_layout()
    if "lc" in which:
        golden_lc()
    if "anchor" in which:
        golden_anchor16()
    if "anchor34" in which:
        golden_anchor34()
