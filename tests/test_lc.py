"""SURVEY.md section 8 f3 -- the LC downstream classifier (eval/model_3d_lc.py, eval/test.py), CPU tier:
the oracle restatement against fixtures produced by the reference's own LC class (tests/golden/lc.npz, lc_layout.json),
the module / engine state_dict layout, and the LCEngine kernel schedule (host simulator, width-reduced net) vs the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from dpc_amd import _lib as L
from dpc_amd.lc import LC, LCEngine, lc_state_dict_keys
from oracle import dpc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDTHS = (8, 16, 32, 32)
TOL = 5e-4


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    return L.load_emulator()


def test_oracle_vs_reference_lc(golden_dir):
    g = np.load(os.path.join(golden_dir, "lc.npz"), allow_pickle=False)
    p = O.make_lc_params_pcg("resnet18", 101)
    x = O.make_input_pcg(2, 8, 5, 64)
    with torch.no_grad():
        out, ctx, _ = O.lc_forward(p, x, "resnet18", train=False)
    assert (out - torch.from_numpy(g["eval_output"])).abs().max().item() < TOL
    assert (ctx - torch.from_numpy(g["eval_context"])).abs().max().item() < TOL
    target = torch.tensor([3, 77])
    loss, acc, grads, out_t, ctx_t, new = O.lc_train_step_reference(p, x, target, "resnet18")
    assert (out_t - torch.from_numpy(g["train_output"])).abs().max().item() < TOL
    assert (ctx_t - torch.from_numpy(g["train_context"])).abs().max().item() < TOL
    assert abs(loss.item() - g["train_loss_acc"][0]) < TOL and acc == pytest.approx(float(g["train_loss_acc"][1]))
    names = [str(n) for n in g["param_names"]]
    assert names == list(grads.keys())
    for i, n in enumerate(names):
        assert grads[n].norm().item() == pytest.approx(float(g["grad_norm"][i]), rel=5e-3, abs=1e-7), n
        st = int(g["grad_substride::" + n])
        rs = torch.from_numpy(g["grad_sub::" + n])
        assert ((grads[n].flatten()[::st] - rs).norm() / rs.norm().clamp_min(1e-12)).item() < 2e-2, n
    for k in ("backbone.bn1", "backbone.layer3.0.bn1", "backbone.layer4.1.bn2", "final_bn"):
        assert np.allclose(new[k + ".running_mean"].numpy(), g["rm::" + k], rtol=1e-4, atol=1e-5), k
        assert np.allclose(new[k + ".running_var"].numpy(), g["rv::" + k], rtol=1e-4, atol=1e-5), k
        assert int(new[k + ".num_batches_tracked"]) == int(g["nbt::" + k]) == 1


def test_state_dict_layout_matches_reference(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "lc_layout.json")))
    ref_keys = [k for k, _, _ in lay["resnet18_101"]]
    assert lc_state_dict_keys("resnet18", 101) == ref_keys == O.lc_state_dict_keys("resnet18", 101)
    m = LC(sample_size=64, num_seq=8, seq_len=5, network="resnet18", dropout=0.5, num_class=101)
    sd = m.state_dict()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == lay["resnet18_101"]
    assert [k for k, _ in m.named_parameters()] == lay["param_order"]
    m.load_state_dict(O.make_lc_params_pcg("resnet18", 101), strict=True)
    with pytest.raises(L.DpcError):
        m(torch.zeros(1, 8, 3, 5, 64, 64))  # CPU tensors: no fallback


def _engine(emu, dtype=torch.float32, B=2, NC=11):
    eng = LCEngine("resnet18", 64, 8, 5, B, "cpu", dtype, WIDTHS, lib=emu, num_class=NC)
    p = O.make_lc_params_pcg("resnet18", NC, WIDTHS)
    eng.load_params(p)
    return eng, p


def test_engine_eval_and_train_vs_oracle(emu):
    B, NC = 2, 11
    eng, p = _engine(emu, B=B, NC=NC)
    assert list(eng.state_dict().keys()) == O.lc_state_dict_keys("resnet18", NC, WIDTHS)
    x = O.make_input_pcg(B, 8, 5, 64)
    target = torch.tensor([3, 7])
    # ---- eval: running statistics everywhere, no dropout
    out, ctx = eng.forward(x, target, train=False)
    with torch.no_grad():
        ro, rc, _ = O.lc_forward(p, x, "resnet18", train=False)
    assert (out - ro).abs().max().item() < 1e-3 and (ctx - rc).abs().max().item() < 1e-3
    rl, ra = O.lc_loss_acc(ro, target)
    assert abs(eng.result[0].item() - rl.item()) < 1e-4 and eng.result[1].item() == pytest.approx(ra.item())
    assert int(eng.BUF["backbone.bn1.num_batches_tracked"]) == 0
    with pytest.raises(L.DpcError):
        eng.backward()
    # ---- train: batch statistics, injected dropout masks shared with the oracle
    g = torch.Generator().manual_seed(3)
    ls, D = eng.last_size, eng.D
    keep = (torch.rand(8, B, ls, ls, D, generator=g) > 0.1).float() / 0.9
    fc_keep = (torch.rand(B, D, generator=g) > 0.5).float() / 0.5
    masks_ref = [keep[i].permute(0, 3, 1, 2).contiguous() for i in range(8)]
    loss, acc, grads, ro, rc, new = O.lc_train_step_reference(p, x, target, "resnet18", masks_ref, fc_keep)
    out, ctx = eng.forward(x, target, train=True, gru_masks=keep.reshape(8, eng.M, D), fc_mask=fc_keep)
    assert (out - ro).abs().max().item() < 1e-3 and (ctx - rc).abs().max().item() < 1e-3
    assert abs(eng.result[0].item() - loss.item()) < 1e-4 and eng.result[1].item() == pytest.approx(acc)
    eng.backward()
    for k, r in grads.items():
        e = (eng.G[k] - r).abs().max().item() / max(r.abs().max().item(), 1e-8)
        assert e < 2e-3, (k, e)
    for k, v in new.items():
        if v.dtype.is_floating_point:
            assert torch.allclose(eng.BUF[k], v, rtol=1e-4, atol=1e-5), k
        else:
            assert int(eng.BUF[k]) == int(v) == 1, k
    # ---- fused Adam on the LC arena (the Philox form of both dropouts is covered at kernel level: test_lc_head_philox)
    before = eng.flat_p.clone()
    eng.adam_step()
    assert eng.step_count == 1 and not torch.equal(before, eng.flat_p)


def test_lc_head_philox(emu):
    """head with in-kernel dropout: keep rate 1 - p, masks change with the optimizer step, backward consistent with forward"""
    import ctypes as C
    B, SQ, D, NC = 64, 4, 32, 11
    g = torch.Generator().manual_seed(1)
    h = torch.randn(B * SQ, D, generator=g)
    bufs = {n: torch.zeros(B, D) for n in ("ctx", "xhat", "bn_out", "y", "dctx")}
    t = dict(bn_weight=torch.ones(D), bn_bias=torch.zeros(D), bn_running_mean=torch.zeros(D), bn_running_var=torch.ones(D),
             bn_num_batches=torch.zeros((), dtype=torch.int64), fc_weight=torch.randn(NC, D, generator=g) * 0.2, fc_bias=torch.zeros(NC),
             target=torch.arange(B) % NC, stat=torch.zeros(2, D), logits=torch.zeros(B, NC), dlogits=torch.zeros(B, NC),
             row_ws=torch.zeros(B, 2), result=torch.zeros(2), g_fc_weight=torch.zeros(NC, D), g_fc_bias=torch.zeros(NC),
             g_bn_weight=torch.zeros(D), g_bn_bias=torch.zeros(D), d_hlast=torch.zeros(B * SQ, D), h_last=h, **bufs)
    step = torch.tensor([4], dtype=torch.int32)
    d = L.LcHeadDesc()
    d.dtype, d.B, d.SQ, d.D, d.num_class, d.train = L.F32, B, SQ, D, NC, 1
    d.p_drop, d.momentum, d.eps, d.seed = 0.5, 0.1, 1e-5, 667
    d.step_dev = step.data_ptr()
    for k, v in t.items():
        setattr(d, k, v.data_ptr())
    emu.call("dpc_lc_head_fwd", C.byref(d), emu.stream())
    y4 = t["y"].clone()
    keep = (y4 != 0) | (t["bn_out"] == 0)
    assert abs(keep.float().mean().item() - 0.5) < 0.05
    assert torch.allclose(y4[keep], (t["bn_out"] * 2.0)[keep])
    emu.call("dpc_lc_head_bwd", C.byref(d), emu.stream())
    assert torch.isfinite(t["d_hlast"]).all() and t["g_fc_weight"].abs().max().item() > 0
    step.fill_(5)
    emu.call("dpc_lc_head_fwd", C.byref(d), emu.stream())
    assert not torch.equal(y4 != 0, t["y"] != 0) and int(t["bn_num_batches"]) == 2


def test_engine_bf16_runs(emu):
    eng, p = _engine(emu, torch.bfloat16)
    x = O.make_input_pcg(2, 8, 5, 64)
    target = torch.tensor([1, 4])
    out, _ = eng.forward(x, target, train=False)
    with torch.no_grad():
        ro, _, _ = O.lc_forward(p, x, "resnet18", train=False)
    assert (out - ro).abs().max().item() < 0.15 * max(ro.abs().max().item(), 1.0)
    assert torch.isfinite(eng.result).all()


def test_module_parameters_alias_the_engine(emu):
    """LC module <-> engine: one storage.  load_state_dict after the first forward reaches the kernels; what the engine's
    train step updates (weights, BatchNorm running buffers, num_batches_tracked) is what state_dict() returns."""
    from dpc_amd.lc import LC
    m = LC(64, 8, 5, "resnet18", 0.5, 11, widths=WIDTHS, _simulator=emu)
    x = O.make_input_pcg(2, 8, 5, 64)
    m.eval()
    out0, _ = m(x)
    eng = m.engine
    sd = m.state_dict()
    assert sd["backbone.conv1.weight"].data_ptr() == eng.PRM["backbone.conv1.weight"].data_ptr()
    assert sd["backbone.bn1.running_mean"].data_ptr() == eng.BUF["backbone.bn1.running_mean"].data_ptr()
    assert sd["agg.cell_list.0.out_gate.weight"].data_ptr() == eng.PRM["agg.ConvGRUCell_00.out_gate.weight"].data_ptr()
    # (1) module -> engine: a new state_dict is seen by the next forward without rebuilding the engine
    new = {k: (v * 0.5 if v.dtype.is_floating_point and k.endswith("final_fc.1.weight") else v.clone()) for k, v in sd.items()}
    m.load_state_dict(new, strict=True)
    out1, _ = m(x)
    assert m.engine is eng and not torch.equal(out0, out1)
    # (2) engine -> module: a train step moves the weights and the running statistics the module reports
    before = {k: v.clone() for k, v in m.state_dict().items()}
    eng.train_step(x, torch.tensor([1, 3]))
    after = m.state_dict()
    assert not torch.equal(before["backbone.layer1.0.conv1.weight"], after["backbone.layer1.0.conv1.weight"])
    assert not torch.equal(before["backbone.bn1.running_mean"], after["backbone.bn1.running_mean"])
    assert int(after["backbone.bn1.num_batches_tracked"]) == int(before["backbone.bn1.num_batches_tracked"]) + 1
    assert int(after["final_bn.num_batches_tracked"]) == int(before["final_bn.num_batches_tracked"]) + 1
    # (3) copies: deepcopy = same values (running statistics included), own storage, own engine at its first forward; a DataParallel
    # replica refuses; pickling the whole module points at state_dict()
    import copy
    import pickle
    c = copy.deepcopy(m)
    assert c.engine is None and not c.training
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), c.state_dict().items()):
        assert ka == kb and torch.equal(va, vb) and (va.numel() == 0 or va.data_ptr() != vb.data_ptr()), ka
    with pytest.raises(RuntimeError, match=r"python -m dpc_amd\.lc_main --gpu"):
        m._replicate_for_data_parallel()(x)
    with pytest.raises(TypeError, match="state_dict"):
        pickle.dumps(m)


def test_lr_schedule_and_train_what():
    """MultiStepLR_Restart_Multiplier (eval/test.py:408-423): the docstring's own table for step=[10,15,20], repeat=3, and the
    schedules main() selects (eval/test.py:93-99)"""
    from dpc_amd.lc_main import lr_milestones, lr_multiplier
    tab = {0: 1, 9: 1, 10: 0.1, 14: 0.1, 15: 0.01, 19: 0.01, 20: 1, 29: 1, 30: 0.1, 35: 0.01, 40: 1, 50: 0.1, 59: 0.01,
           60: 0.01, 61: 0.01, 75: 0.01, 1000: 0.01}
    for ep, want in tab.items():
        assert lr_multiplier(ep, 0.1, [10, 15, 20], 3) == pytest.approx(want), ep
    assert lr_milestones("ucf101", 128) == [60, 80, 100] and lr_milestones("ucf101", 224) == [300, 400, 500]
    assert lr_milestones("hmdb51", 128) == [150, 250, 300]
    assert [lr_multiplier(e, 0.1, [60, 80, 100], 1) for e in (0, 59, 60, 80, 99, 100, 500)] == pytest.approx([1, 1, .1, .01, .01, .01, .01])
