"""The CPU oracle (oracle/dpc_oracle.py) against fixtures the reference itself
produced (tests/golden/make_golden.py).  Tolerances: the reference's own fp32
noise floor is ~1e-4 (fp32 vs fp64, oneDNN on/off), so scores
are held to 5e-4 absolute; integer/byte results are bit-exact."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dpc_oracle as O

TOL = 5e-4


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("B,P,SQ", [(4, 3, 16), (3, 5, 49), (2, 1, 4)])
def test_mask_bit_exact(golden_dir, B, P, SQ):
    g = load(golden_dir, "mask.npz")
    key = f"{B}_{P}_{SQ}"
    m = O.mask_closed_form(B, P, SQ)
    sha = np.frombuffer(hashlib.sha256(m.numpy().tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(sha, g["sha_" + key])
    vals, counts = np.unique(m.numpy(), return_counts=True)
    assert np.array_equal(vals, g["vals_" + key]) and np.array_equal(counts, g["counts_" + key])
    assert np.array_equal(O.target_from_mask(m).numpy(), g["target_" + key])
    assert np.array_equal(g["target_" + key], np.arange(B * P * SQ))  # closed form: arange
    if (B, P, SQ) == (2, 1, 4):
        assert np.array_equal(m.numpy(), g["mask_" + key])
    if B * P * SQ <= 200:
        assert torch.equal(O.mask_by_reference_steps(B, P, SQ), m)


@pytest.mark.parametrize("tag,net,size,B", [("r18_64_b2", "resnet18", 64, 2), ("r34_64_b2", "resnet34", 64, 2),
                                            ("r18_128_b4", "resnet18", 128, 4)])
def test_eval_score(golden_dir, tag, net, size, B):
    g = load(golden_dir, "eval_scores.npz")
    p = O.make_params_pcg(net)
    x = O.make_input_pcg(B, 8, 5, size)
    with torch.no_grad():
        score = O.dpc_forward(p, x, net)
    ref = torch.from_numpy(g["score_" + tag])
    assert score.shape == ref.shape
    err = (score - ref).abs().max().item()
    assert err < TOL, err
    if tag == "r18_64_b2":
        loss, accs = O.loss_and_topk(score)
        e = g["evalloss_" + tag]
        assert abs(loss.item() - e[0]) < TOL
        assert [a.item() for a in accs] == pytest.approx(list(e[1:]), abs=1e-7)


def test_train_step_p0(golden_dir):
    g = load(golden_dir, "train.npz")
    p = O.make_params_pcg("resnet18")
    x = O.make_input_pcg(2, 8, 5, 64)
    loss, accs, grads, score = O.train_step_reference(p, x, "resnet18", 3, dropout_masks=None)
    assert (score - torch.from_numpy(g["score_p0"])).abs().max().item() < TOL
    e = g["loss_topk_p0"]
    assert abs(loss.item() - e[0]) < TOL
    assert accs == pytest.approx(list(e[1:]), abs=1e-7)
    names = [str(n) for n in g["param_names"]]
    assert names == O.unique_param_names(p)
    for i, n in enumerate(names):
        gn = grads[n].norm().item()
        assert gn == pytest.approx(float(g["grad_norm_p0"][i]), rel=2e-3, abs=1e-6), n
        if grads[n].numel() >= 8:
            ref = g["grad_head_p0"][i]
            assert np.allclose(grads[n].flatten()[:8].numpy(), ref, rtol=5e-3, atol=2e-4 * max(gn, 1e-3)), n
    for k in g.files:
        if k.startswith("grad_sub_p0::"):
            n = k.split("::", 1)[1]
            stride = int(g["grad_substride_p0::" + n])
            mine = grads[n].flatten()[::stride].numpy()
            scale = np.abs(g[k]).max()
            assert np.abs(mine - g[k]).max() < 2e-3 * scale + 1e-7, n
    # one Adam step (main.py:80-81)
    for i, n in enumerate(names):
        w = p[n].clone()
        O.adam_step(w, grads[n], torch.zeros_like(w), torch.zeros_like(w), 1)
        # first Adam step moves every element by ~lr*sign(g): an element whose grad is at
        # the fp32 noise floor may flip sign (2*lr each); allow ~1e-5 of the elements to flip
        slack = 2e-3 * max(1.0, 2e-5 * w.numel())
        assert w.double().sum().item() == pytest.approx(float(g["adam_sum_p0"][i]), rel=1e-5, abs=slack), n
        if w.numel() >= 8:
            # sign(g)*lr steps: elements whose grad is ~0 may flip; compare with slack of 2*lr
            assert np.abs(w.flatten()[:8].numpy() - g["adam_head_p0"][i]).max() < 2.1e-3, n


def test_injected_dropout(golden_dir):
    g = load(golden_dir, "train.npz")
    shape = tuple(int(v) for v in g["drop_keep_shape"])
    keep = np.unpackbits(g["drop_keep_bits"])[: int(np.prod(shape))].reshape(shape)
    masks = [torch.from_numpy(keep[i].astype(np.float32)) / 0.9 for i in range(shape[0])]
    p = O.make_params_pcg("resnet18")
    x = O.make_input_pcg(2, 8, 5, 64)
    with torch.no_grad():
        score = O.dpc_forward(p, x, "resnet18", 3, dropout_masks=masks)
    assert (score - torch.from_numpy(g["score_drop"])).abs().max().item() < TOL


def _block_params(g, tag):
    pre = tag + "::w::"
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("tag,is3d,stride,final_relu", [("bb2d_s2_ds", False, 2, True), ("bb2d_s1", False, 1, True),
                                                       ("bb3d_s2_ds", True, 2, True), ("bb3d_s1_norelu", True, 1, False)])
def test_basic_block_fwd_bwd(golden_dir, tag, is3d, stride, final_relu):
    g = load(golden_dir, "ops.npz")
    w = {"blk." + k: v.clone().requires_grad_(True) for k, v in _block_params(g, tag).items()}
    x = torch.from_numpy(g[tag + "::x"]).requires_grad_(True)
    y = O.basic_block(x, w, "blk.", is3d, stride, final_relu)
    assert (y - torch.from_numpy(g[tag + "::y"])).abs().max().item() < 1e-4
    y.backward(torch.from_numpy(g[tag + "::gy"]))
    assert (x.grad - torch.from_numpy(g[tag + "::gx"])).abs().max().item() < 1e-3
    for k, v in w.items():
        ref = torch.from_numpy(g[f"{tag}::gw::{k[4:]}"])
        assert (v.grad - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item()), k


def test_gru_cell(golden_dir):
    g = load(golden_dir, "ops.npz")
    w = {"c." + k: v.clone().requires_grad_(True) for k, v in _block_params(g, "gru").items()}
    x = torch.from_numpy(g["gru::x"]).requires_grad_(True)
    h = torch.from_numpy(g["gru::h"]).requires_grad_(True)
    hn = O.convgru_cell(x, h, w, "c.")
    assert (hn - torch.from_numpy(g["gru::hn"])).abs().max().item() < 1e-5
    hn.backward(torch.from_numpy(g["gru::gh"]))
    assert (x.grad - torch.from_numpy(g["gru::gx"])).abs().max().item() < 1e-5
    assert (h.grad - torch.from_numpy(g["gru::ghprev"])).abs().max().item() < 1e-5
    for k, v in w.items():
        assert (v.grad - torch.from_numpy(g["gru::gw::" + k[2:]])).abs().max().item() < 1e-4, k


def test_topk_and_loss(golden_dir):
    g = load(golden_dir, "ops.npz")
    sc = torch.from_numpy(g["topk::score"]).view(2, 3, 4, 2, 3, 4)
    loss, accs = O.loss_and_topk(sc)
    assert abs(loss.item() - float(g["topk::loss"])) < 1e-6
    assert [a.item() for a in accs] == pytest.approx(list(g["topk::acc"]), abs=1e-7)


def test_eval_score_r34_pred_step5(golden_dir):
    """cfg5's combination (resnet34 + pred_step 5): oracle vs the reference's own score / loss / top-k"""
    g = load(golden_dir, "eval_scores_p5.npz")
    p = O.make_params_pcg("resnet34")
    x = O.make_input_pcg(2, 8, 5, 64)
    with torch.no_grad():
        score = O.dpc_forward(p, x, "resnet34", 5)
    ref = torch.from_numpy(g["score_r34_64_b2_p5"])
    assert score.shape == ref.shape == (2, 5, 4, 2, 5, 4)
    assert (score - ref).abs().max().item() < TOL
    loss, accs = O.loss_and_topk(score)
    e = g["evalloss_r34_64_b2_p5"]
    assert abs(loss.item() - e[0]) < TOL
    assert [a.item() for a in accs] == pytest.approx(list(e[1:]), abs=1e-7)


def test_anchor_r18_128_b16(golden_dir):
    """the fixture that anchors the bf16 throughput mode (tests/test_engine_gpu.py): the oracle reproduces it in fp32"""
    g = load(golden_dir, "anchor_r18_128_b16.npz")
    p = O.make_params_pcg("resnet18")
    x = O.make_input_pcg(16, 8, 5, 128)
    loss, accs, grads, score = O.train_step_reference(p, x, "resnet18", 3, dropout_masks=None)
    stride = int(g["score_stride"])
    ref = torch.from_numpy(g["score_sub"])
    assert (score.flatten()[::stride] - ref).abs().max().item() < TOL
    assert abs(score.norm().item() / float(g["score_norm"]) - 1) < 1e-5
    e = g["loss_topk"]
    assert abs(loss.item() - e[0]) < TOL and accs == pytest.approx(list(e[1:]), abs=1e-7)
    names = [str(n) for n in g["param_names"]]
    assert names == O.unique_param_names(p)
    for i, n in enumerate(names):
        assert grads[n].norm().item() == pytest.approx(float(g["grad_norm"][i]), rel=5e-3, abs=1e-7), n
        st = int(g["grad_substride::" + n])
        rs = torch.from_numpy(g["grad_sub::" + n])
        err = ((grads[n].flatten()[::st] - rs).norm() / rs.norm().clamp_min(1e-12)).item()
        assert err < 2e-2, (n, err)  # same fp32 arithmetic, different summation order + ReLU-boundary flips
