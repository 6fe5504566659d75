"""CPU tier, end to end: the engine's full kernel schedule (forward, CE/top-k, backward,
Adam) executed by the host SIMT simulator on a width-reduced 2d3d-ResNet18 (the reference
block classes are width-parametric, backbone/resnet_2d3d.py:50,86) against the oracle."""
import os
import subprocess

import pytest
import torch

from dpc_amd import _lib as L
from dpc_amd.engine import DPCEngine, param_shapes
from oracle import dpc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDTHS = (8, 16, 32, 32)


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    return L.load_emulator()


def test_param_inventory_matches_reference_keys():
    for net in ("resnet18", "resnet34"):
        mine = param_shapes(net)
        ref = O.param_shapes(net, with_alias=False)
        assert list(mine.items()) == list(ref.items())
    with pytest.raises(IOError):
        param_shapes("resnet50")


def test_engine_refuses_cpu_without_simulator():
    with pytest.raises(L.DpcError):
        DPCEngine("resnet18", 64, 8, 5, 3, 2, "cpu", torch.float32, WIDTHS)


def test_train_step_parity_f32(emu):
    B, size = 2, 64
    eng = DPCEngine("resnet18", size, 8, 5, 3, B, "cpu", torch.float32, WIDTHS, lib=emu)
    p = O.make_params_pcg("resnet18", WIDTHS)
    eng.load_params(p)
    x = O.make_input_pcg(B, 8, 5, size)
    # injected dropout masks: [n_steps, M, D] pre-scaled keep masks, shared with the oracle
    g = torch.Generator().manual_seed(5)
    keep = (torch.rand(eng.n_steps, B, eng.last_size, eng.last_size, eng.D, generator=g) > 0.1).float() / 0.9
    masks_eng = keep.reshape(eng.n_steps, eng.M, eng.D)
    masks_ref = [keep[i].permute(0, 3, 1, 2).contiguous() for i in range(eng.n_steps)]
    masks_ref.append(torch.ones_like(masks_ref[0]))  # the reference's last GRU step is dead code (model_3d.py:70-72)
    loss, accs, grads, ref = O.train_step_reference(p, x, "resnet18", 3, masks_ref)
    score = eng.forward(x, train=True, dropout_masks=masks_eng)
    assert (score - ref).abs().max().item() < 1e-3  # north_star tolerance, fp32
    res = eng.loss_topk(True)
    assert abs(res[0].item() - loss.item()) < 1e-4
    assert res[1:].tolist() == pytest.approx(accs, abs=1e-6)
    eng.backward()
    for k, r in grads.items():
        e = (eng.G[k] - r).abs().max().item() / max(r.abs().max().item(), 1e-8)
        assert e < 1e-3, (k, e)
    # mask + Adam
    assert torch.equal(eng.get_mask(), O.mask_closed_form(B, 3, eng.SQ))
    before = {k: v.clone() for k, v in eng.PRM.items()}
    eng.adam_step()
    for k in ("backbone.layer2.0.conv1.weight", "agg.ConvGRUCell_00.out_gate.weight", "network_pred.2.bias"):
        w = before[k].clone()
        O.adam_step(w, grads[k], torch.zeros_like(w), torch.zeros_like(w), 1)
        assert (eng.PRM[k] - w).abs().max().item() < 2.1e-3  # sign flips of ~0 grads move 2*lr
        assert ((eng.PRM[k] - w).abs() > 1e-5).float().mean().item() < 0.01


def test_forward_bf16_smoke(emu):
    B, size = 2, 64
    eng = DPCEngine("resnet18", size, 8, 5, 3, B, "cpu", torch.bfloat16, WIDTHS, lib=emu)
    p = O.make_params_pcg("resnet18", WIDTHS)
    eng.load_params(p)
    x = O.make_input_pcg(B, 8, 5, size)
    score = eng.forward(x, train=False)
    with torch.no_grad():
        ref = O.dpc_forward(p, x, "resnet18", 3)
    # bf16 operands: not a parity claim, only that the throughput mode computes the same function
    assert (score - ref).abs().max().item() < 0.15 * ref.abs().max().item()
    res = eng.loss_topk(True)
    eng.backward()
    assert torch.isfinite(eng.flat_g).all() and torch.isfinite(res).all()


def test_fused_score_path_matches_materialised(emu):
    """throughput mode: forward(materialise=False) + fused loss + fused score backward (no [R][R] tensor) against the
    materialised path of the same engine on the same step (same Philox dropout masks: new_draw=False repeats the draw)"""
    B, size = 1, 64
    eng = DPCEngine("resnet18", size, 8, 5, 3, B, "cpu", torch.bfloat16, WIDTHS, lib=emu, score_path="fused")
    assert eng.score_fusable
    eng.load_params(O.make_params_pcg("resnet18", WIDTHS))
    x = O.make_input_pcg(B, 8, 5, size)
    score = eng.forward(x, train=True, materialise=True)
    assert score is not None and eng.score_mode == "materialised"
    res_m = eng.loss_topk(True).clone()
    eng.backward()
    g_m, dp_m, df_m = eng.flat_g.clone(), eng.d_pred.clone(), eng.d_finf.clone()
    assert eng.forward(x, train=True, materialise=False, new_draw=False) is None and eng.score_mode == "fused"
    res_f = eng.loss_topk(True).clone()
    eng.backward()
    assert abs(res_f[0].item() - res_m[0].item()) < 1e-4
    assert res_f[1:].tolist() == pytest.approx(res_m[1:].tolist(), abs=1.01 / eng.R)
    for a, b in ((eng.d_pred, dp_m), (eng.d_finf, df_m)):
        assert ((a - b).norm() / b.norm()).item() < 2e-2
    assert ((eng.flat_g - g_m).norm() / g_m.norm()).item() < 3e-2
    with pytest.raises(L.DpcError):
        eng.backward(dscore_external=torch.zeros(eng.R, eng.R))


def test_module_train_mode_draws_fresh_dropout_masks(emu):
    """nn.Module boundary with an EXTERNAL optimizer (adam_step never runs): every train-mode forward must still draw new
    ConvGRU dropout masks (backbone/convrnn.py:78 draws per call); eval forwards draw none and agree with each other."""
    from dpc_amd.model import DPC_RNN
    m = DPC_RNN(64, 8, 5, 3, "resnet18", widths=WIDTHS, _simulator=emu)
    x = O.make_input_pcg(1, 8, 5, 64)
    m.train()
    s1 = m(x)[0].detach().clone()
    k1 = m.engine.dropout_masks_of_step().clone()
    s2 = m(x)[0].detach().clone()
    k2 = m.engine.dropout_masks_of_step().clone()
    assert int(m.engine.dev_draw.item()) == 2 and m.engine.step_count == 0
    assert not torch.equal(k1, k2) and not torch.equal(s1, s2)
    assert abs(k2.ne(0).float().mean().item() - 0.9) < 0.03
    # the backward of a forward regenerates THAT forward's bits: gradient of the second call == injected-mask run of the same masks
    s3 = m(x)[0]
    k3 = m.engine.dropout_masks_of_step().clone()
    s3.sum().backward()
    g_philox = m.network_pred[2].weight.grad.clone() if hasattr(m.network_pred, "__getitem__") else dict(m.named_parameters())["network_pred.2.weight"].grad.clone()
    eng = m.engine
    sc = eng.forward(x, train=True, dropout_masks=k3)
    assert torch.equal(sc.reshape(-1), s3.detach().reshape(-1))
    eng.backward(dscore_external=torch.ones(eng.R, eng.R))
    assert torch.equal(eng.G["network_pred.2.weight"], g_philox)
    m.eval()
    e1 = m(x)[0].detach().clone()
    e2 = m(x)[0].detach().clone()
    assert torch.equal(e1, e2) and int(m.engine.dev_draw.item()) == 3


def test_fused_backward_epilogues_match_the_separate_kernels(emu):
    """fold=True (default): BatchNorm-backward reductions ride in the producing input-gradient's epilogue and the residual
    branch is gated on the fly (no dz tensor) -- against fold=False, the separate-kernel schedule, on the same step."""
    B, size = 1, 64
    x = O.make_input_pcg(B, 8, 5, size)
    out = []
    for fold in (False, True):
        eng = DPCEngine("resnet18", size, 8, 5, 3, B, "cpu", torch.float32, WIDTHS, lib=emu, fold=fold)
        eng.load_params(O.make_params_pcg("resnet18", WIDTHS))
        flags = [(b.fold_c1, b.gate, b.fold_prev) for b in eng.blocks]
        assert all(f[0] == fold for f in flags)                                 # conv2's input-gradient always folds bn1
        assert [f[1] for f in flags] == [fold and b.ds is None for b in eng.blocks]   # downsample blocks accumulate in place instead
        assert [f[2] for f in flags] == [fold and b.ds is None and i > 0 for i, b in enumerate(eng.blocks)]
        eng.forward(x, train=False)
        eng.loss_topk(True)
        eng.backward()
        out.append(eng.flat_g.clone())
    # same arithmetic on the same stored values; only the summation order of the partial sums differs
    for k, (o, n) in eng.offsets.items():
        a, b = out[0][o:o + n], out[1][o:o + n]
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), k


def test_module_boundary_copies_nothing_and_keeps_torch_semantics(emu):
    """dpc/main.py:198-231 over dpc_amd.model.DPC_RNN: the returned score is the engine's buffer, after loss.backward() every .grad is
    a view of the gradient arena (AccumulateGrad adopts the fresh views), dpc_amd.optim.Adam == torch.optim.Adam step for step,
    a second backward without zero_grad accumulates like torch, and the optimizer state moves to torch's Adam and back."""
    from dpc_amd.model import DPC_RNN
    from dpc_amd.optim import Adam
    x = O.make_input_pcg(1, 8, 5, 64)
    crit = torch.nn.CrossEntropyLoss()

    def make(opt_cls):
        m = DPC_RNN(64, 8, 5, 3, "resnet18", widths=WIDTHS, seed=4, _simulator=emu)
        m.eval()   # no dropout: the two replicas see the same function
        return m, opt_cls(m.parameters(), lr=1e-3, weight_decay=1e-5)

    def loss_of(m):
        score_, mask_ = m(x)
        B, NP, SQ, B2, NS, _ = mask_.size()
        target = (mask_ == 1).view(B * NP * SQ, B2 * NS * SQ).to(int).argmax(dim=1)
        return crit(score_.view(B * NP * SQ, B2 * NS * SQ), target), score_

    (ma, oa), (mb, ob) = make(Adam), make(torch.optim.Adam)
    with pytest.raises(RuntimeError):
        oa.step()   # no engine yet: nothing to update, loud

    def backward_once(m, o, zero=True):
        loss, score_ = loss_of(m)
        assert score_.data_ptr() in [b.data_ptr() for b in m._score_bufs]
        if zero:
            o.zero_grad()
        loss.backward()
        adopted = [p.grad.data_ptr() == m.engine.G[k].data_ptr() for k, p in m.named_parameters() if not k.startswith("agg.cell_list")]
        assert all(adopted), f"{sum(adopted)} of {len(adopted)} gradients are arena views"

    backward_once(ma, oa)
    g1 = {k: p.grad.clone() for k, p in ma.named_parameters()}
    backward_once(ma, oa, zero=False)   # no zero_grad: torch accumulates -- same input, same parameters: exactly twice the gradient
    for k, p in ma.named_parameters():
        assert torch.equal(p.grad, 2 * g1[k]), k
        p.grad.mul_(0.5)
    backward_once(mb, ob)
    for k, p in mb.named_parameters():
        assert torch.equal(p.grad, g1[k]), k
    oa.step()
    ob.step()
    assert ma.engine.step_count == 1
    for (ka, pa), (kb, pb) in zip(ma.named_parameters(), mb.named_parameters()):   # fused arena Adam == torch.optim.Adam (1 ulp)
        assert ka == kb and (pa - pb).abs().max().item() <= 1e-7 * max(1.0, pb.abs().max().item()), ka
    # optimizer state in torch.optim.Adam's layout: loads into torch's Adam, and into a fresh dpc_amd.optim.Adam (at once when the
    # engine exists, at the first step when it does not)
    sd = oa.state_dict()
    ob.load_state_dict(sd)
    assert float(ob.state_dict()["state"][0]["step"]) == 1.0
    m_before = ma.engine.flat_m.clone()
    ma.engine.flat_m.zero_()
    oa2 = Adam(ma.parameters(), lr=5e-4, weight_decay=0.0)
    oa2.load_state_dict(sd)
    assert torch.equal(ma.engine.flat_m, m_before) and oa2.param_groups[0]["lr"] == 1e-3
    mc, oc = make(Adam)
    oc.load_state_dict(sd)
    assert oc._pending_state is sd
    # compute dtype through the module's own calls; parameters stay f32
    mc.bfloat16()
    assert mc.compute_dtype == torch.bfloat16 and next(mc.parameters()).dtype == torch.float32
    assert mc.float().compute_dtype == torch.float32


def _ref_loop_step(model, opt, x, crit):
    """the loop lines of dpc/main.py:198-231 (forward, target from the mask, CrossEntropyLoss, zero_grad / backward / step)"""
    score_, mask_ = model(x)
    B, NP, SQ, B2, NS, _ = mask_.size()
    target = (mask_ == 1).view(B * NP * SQ, B2 * NS * SQ).to(int).argmax(dim=1)
    loss = crit(score_.view(B * NP * SQ, B2 * NS * SQ), target)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.item()


def _small(emu, seed):
    """the drop-in at half the sequence (4 blocks, 1 prediction step): these tests are about the host side"""
    from dpc_amd.model import DPC_RNN
    return DPC_RNN(64, 4, 5, 1, "resnet18", widths=WIDTHS, seed=seed, _simulator=emu).eval()


def test_data_parallel_entry(emu):
    """dpc/main.py:65-66: `model = nn.DataParallel(model)`.  With one (or no) visible device DataParallel calls the wrapped module
    itself (torch nn/parallel/data_parallel.py:187-195): the wrapped drop-in steps exactly like the bare one.  (Several devices:
    test_data_parallel_replicas below.)"""
    from dpc_amd.optim import Adam
    x = O.make_input_pcg(1, 4, 5, 64)
    crit = torch.nn.CrossEntropyLoss()
    bare = _small(emu, 4)
    wrapped = torch.nn.DataParallel(_small(emu, 4)).eval()
    assert list(wrapped.state_dict()) == ["module." + k for k in bare.state_dict()]   # the keys dpc/main.py:170 saves
    la = _ref_loop_step(bare, Adam(bare.parameters(), lr=1e-3, weight_decay=1e-5), x, crit)
    lb = _ref_loop_step(wrapped, Adam(wrapped.parameters(), lr=1e-3, weight_decay=1e-5), x, crit)
    assert la == lb
    for (ka, pa), (kb, pb) in zip(bare.named_parameters(), wrapped.module.named_parameters()):
        assert ka == kb and torch.equal(pa, pb), ka


def _cpu_replicas(root, n):
    """what torch.nn.parallel.replicate builds per forward (replicate.py): shallow module copies (_replicate_for_data_parallel) wired
    like the original, whose parameters are NON-LEAF copies set as plain attributes.  p.clone() stands in for Broadcast.apply: a
    differentiable copy whose gradients add up on the root's leaf parameter, as Broadcast.backward's reduce-add does across devices."""
    modules = list(root.modules())
    idx = {m: i for i, m in enumerate(modules)}
    out = []
    for j in range(n):
        copies = [m._replicate_for_data_parallel() for m in modules]
        for i, m in enumerate(modules):
            for key, child in m._modules.items():
                setattr(copies[i], key, copies[idx[child]])
            for key, prm in m._parameters.items():
                setattr(copies[i], key, prm.clone())
        copies[0]._replica_slot = j          # on the GPU the device index tells the replicas apart
        out.append(copies[0])
    return out


def test_data_parallel_replicas(emu):
    """dpc/main.py:65-66 with several ids in --gpu: DataParallel scatters the batch, replicates the module, runs the replicas and
    gathers [score, mask] along dim 0 (data_parallel.py:187-198).  A replica of the drop-in finds the engine of ITS device in the
    root's registry, loads the broadcast parameters, and its backward feeds the broadcast copies: the root's .grad is the gradient
    of the reference's loss over the gathered rows = the mean of the per-shard reference gradients (per-GPU negatives and batch
    statistics, dpc/main.py:180,211-213).  Two hand-made replicas on the simulator; the GPU form needs two devices."""
    import warnings
    root = _small(emu, 4)
    xg = O.make_input_pcg(2, 4, 5, 64)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(root.parameters(), lr=1e-3, weight_decay=1e-5)      # dpc/main.py:80-81 (root parameters stay ordinary leaves)
    p0 = {k: v.detach().clone() for k, v in root.named_parameters()}
    engines = []
    for step in range(2):
        reps = _cpu_replicas(root, 2)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            outs = [r(xg[j:j + 1]) for j, r in enumerate(reps)]                 # parallel_apply
        assert (len(w) == 1 and "dpc_amd.main --gpu" in str(w[0].message)) == (step == 0)   # the pointer to the fast route, once
        score_, mask_ = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])   # gather
        B, NP, SQ, B2, NS, _ = mask_.size()                                   # dpc/main.py:209-217: B = 2 replicas x B2 = 1
        assert (B, B2) == (2, 1)
        target = (mask_ == 1).view(B * NP * SQ, B2 * NS * SQ).to(int).argmax(dim=1)
        loss = crit(score_.view(B * NP * SQ, B2 * NS * SQ), target)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            grads = {k: v.grad.clone() for k, v in root.named_parameters()}
            loss0 = loss.item()
        opt.step()
        engines.append(sorted(id(e) for e in root._replica_engines.values()))
    assert len(root._replica_engines) == 2 and engines[0] == engines[1]          # one engine per device, kept across steps
    assert sorted(e.seed for e in root._replica_engines.values()) == [233, 234]  # per-replica dropout streams
    assert root.engine is None                                                  # the root itself never ran
    want_loss, want = 0.0, {k: torch.zeros_like(v) for k, v in p0.items()}
    names = list(p0)
    for j in range(2):   # the oracle on ONE shard (4 score rows: too few for train_step_reference's top-5)
        leaves = {k: p0[k].clone().requires_grad_(True) for k in names}
        sc = O.dpc_forward(leaves, xg[j:j + 1], "resnet18", 1, None)
        l = torch.nn.functional.cross_entropy(sc.reshape(4, 4), torch.arange(4))
        g = dict(zip(names, torch.autograd.grad(l, [leaves[k] for k in names])))
        want_loss += 0.5 * l.item()
        for k in want:
            want[k] += 0.5 * g[k]
    assert abs(loss0 - want_loss) < 1e-4
    for k, g in want.items():
        assert (grads[k] - g).abs().max().item() < 2e-3 * max(g.abs().max().item(), 1e-6), k
    assert any(not torch.equal(p0[k], v.detach()) for k, v in root.named_parameters())   # torch's Adam stepped the root's leaves


def test_engine_rebuild_keeps_the_optimizer_state(emu):
    """a forward with another batch size rebuilds the engine; dpc_amd.optim.Adam must go on with the moments, step
    counter and bias corrections it had -- as torch.optim.Adam over the same module does (its state is keyed on the Parameters).
    The second step is checked against Adam's recurrences written out: m2 = b1 m1 + (1 - b1) g2 etc. with step-2 bias corrections."""
    from dpc_amd.optim import Adam
    crit = torch.nn.CrossEntropyLoss()
    m = _small(emu, 5)
    opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
    _ref_loop_step(m, opt, O.make_input_pcg(1, 4, 5, 64), crit)
    first = m.engine
    m1, v1, p1 = first.flat_m.clone(), first.flat_v.clone(), first.flat_p.clone()
    assert first.step_count == 1 and m1.abs().max().item() > 0
    _ref_loop_step(m, opt, O.make_input_pcg(2, 4, 5, 64), crit)   # B = 2: a new engine
    eng = m.engine
    assert eng is not first and eng.B == 2 and eng.step_count == 2 and int(eng.dev_step.item()) == 2
    assert float(opt.state_dict()["state"][0]["step"]) == 2.0
    g = eng.flat_g + 1e-5 * p1                     # L2 weight decay goes into the gradient (torch.optim.Adam, dpc/main.py:80)
    m2, v2 = 0.9 * m1 + 0.1 * g, 0.999 * v1 + 0.001 * g * g
    assert torch.allclose(eng.flat_m, m2, rtol=1e-5, atol=1e-9) and torch.allclose(eng.flat_v, v2, rtol=1e-5, atol=1e-12)
    p2 = p1 - 1e-3 * (m2 / (1 - 0.9 ** 2)) / ((v2 / (1 - 0.999 ** 2)).sqrt() + 1e-8)
    # (zeroed moments / a restarted bias correction would move every weight by ~lr = 1e-3 instead)
    assert (eng.flat_p - p2).abs().max().item() <= 2e-6


def test_module_dtype_calls(monkeypatch):
    """parameters are f32 master weights whatever the caller casts; what cannot be honoured raises instead of silently running f32"""
    from dpc_amd.model import DPC_RNN
    m = DPC_RNN(64, 8, 5, 3, "resnet18", widths=WIDTHS)
    assert m.to(torch.bfloat16).compute_dtype == torch.bfloat16 and next(m.parameters()).dtype == torch.float32
    assert m.to(dtype=torch.float32).compute_dtype == torch.float32
    for call in (m.half, m.double, lambda: m.to(torch.float16)):
        with pytest.raises(TypeError):
            call()
    monkeypatch.setenv("DPC_COMPUTE_DTYPE", "fp8")
    with pytest.raises(ValueError, match="accepted values"):
        DPC_RNN(64, 8, 5, 3, "resnet18", widths=WIDTHS)
    monkeypatch.setenv("DPC_COMPUTE_DTYPE", "bf16")
    assert DPC_RNN(64, 8, 5, 3, "resnet18", widths=WIDTHS).compute_dtype == torch.bfloat16


def test_cast_parameters_are_refused(emu):
    from dpc_amd.model import DPC_RNN
    m = DPC_RNN(64, 4, 5, 1, "resnet18", widths=WIDTHS, _simulator=emu)
    torch.nn.Module.double(m)   # what .double() would have done
    with pytest.raises(TypeError, match="float32 master parameters"):
        m(O.make_input_pcg(1, 4, 5, 64))


@pytest.mark.parametrize("N,SL,P,B,size", [(6, 4, 2, 1, 64), (5, 8, 1, 1, 64), (4, 5, 2, 3, 64), (4, 6, 1, 1, 96)])
def test_other_sequence_shapes(emu, N, SL, P, B, size):
    """the reference's constructor takes any num_seq / seq_len / pred_step (dpc/model_3d.py:16-25: last_duration = ceil(seq_len / 4),
    last_size = ceil(sample_size / 32)); BASELINE's configurations only use 8 / 5 / 3|5.  Eval-mode score against the oracle for
    other block counts, block lengths (T path 4 -> 2 -> 1, 8 -> 4 -> 2, 6 -> 3 -> 2), an odd batch and a 3 x 3 feature map."""
    eng = DPCEngine("resnet18", size, N, SL, P, B, "cpu", torch.float32, WIDTHS, lib=emu)
    p = O.make_params_pcg("resnet18", WIDTHS)
    eng.load_params(p)
    x = O.make_input_pcg(B, N, SL, size)
    score = eng.forward(x, train=False)
    ref = O.dpc_forward(p, x, "resnet18", P)
    assert tuple(score.shape) == tuple(ref.shape) == (B, P, eng.SQ, B, P, eng.SQ)
    assert (score - ref).abs().max().item() < 1e-4
    assert torch.equal(eng.get_mask(), O.mask_closed_form(B, P, eng.SQ))


@pytest.mark.parametrize("size,B", [(66, 1), (80, 1)])
def test_image_sizes_that_are_not_multiples_of_32(emu, size, B):
    """last_size = ceil(sample_size / 32) (dpc/model_3d.py:25): 66 -> 33 -> 17 -> 9 -> 5 -> 3 and 80 -> 40 -> 20 -> 10 -> 5 -> 3 -- odd planes
    under the stem's stride, the max-pool and every strided conv.  Score and every gradient against the oracle (f32, simulator)."""
    N, SL, P = 4, 5, 1
    eng = DPCEngine("resnet18", size, N, SL, P, B, "cpu", torch.float32, WIDTHS, lib=emu)
    p = O.make_params_pcg("resnet18", WIDTHS)
    eng.load_params(p)
    x = O.make_input_pcg(B, N, SL, size)
    score = eng.forward(x, train=False)
    eng.loss_topk(True)
    eng.backward()
    loss, accs, grads, ref = O.train_step_reference(p, x, "resnet18", P, None)
    assert eng.SQ == 9 and (score - ref).abs().max().item() < 1e-4
    worst = max(((eng.G[k] - g).norm() / g.norm().clamp_min(1e-12)).item() for k, g in grads.items())
    assert worst < 1e-3, worst


def test_module_copies_and_no_grad(emu):
    """what a user of the reference may do with the module besides the training loop: forward under torch.no_grad() (validate(),
    dpc/main.py:253), copy.deepcopy (an averaged / EMA copy): same values, own engine, independent parameters; pickling the whole module
    is refused with a pointer to state_dict()"""
    import copy
    import pickle
    m = _small(emu, 6)
    x = O.make_input_pcg(1, 4, 5, 64)
    with torch.no_grad():
        s0, _ = m(x)
    assert not s0.requires_grad
    s1, _ = m(x)
    assert s1.requires_grad and torch.equal(s0, s1.detach())
    c = copy.deepcopy(m)
    assert c.engine is None and c is not m and not c.training
    for (ka, pa), (kb, pb) in zip(m.named_parameters(), c.named_parameters()):
        assert ka == kb and torch.equal(pa, pb) and pa.data_ptr() != pb.data_ptr()
    sc, _ = c(x)
    assert c.engine is not m.engine and torch.equal(sc.detach(), s0)
    with torch.no_grad():
        next(c.parameters()).add_(1.0)
    assert not torch.equal(next(c.parameters()), next(m.parameters()))
    with pytest.raises(TypeError, match="state_dict"):
        pickle.dumps(m)
