"""GPU tier: the other BASELINE.json configurations' shapes (resnet34, img_dim 224 => 7x7 feature map,
pred_step 5, ragged score width) at a small batch, f32, against the CPU oracle run in the same process."""
import pytest
import torch

from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LIMIT = 0.25


@pytest.mark.parametrize("net,size,B,P", [("resnet34", 224, 2, 3), ("resnet18", 224, 2, 5), ("resnet18", 96, 3, 2), ("resnet18", 128, 2, 3)])
def test_config_shapes_vs_oracle(net, size, B, P):
    eng = DPCEngine(net, size, 8, 5, P, B, DEV, torch.float32)
    p = O.make_params_pcg(net)
    eng.load_params(p)
    x = O.make_input_pcg(B, 8, 5, size)
    score = eng.forward(x.to(DEV), train=False).cpu()
    res = eng.loss_topk(True).cpu()
    eng.backward()
    torch.cuda.synchronize()
    loss, accs, grads, ref = O.train_step_reference(p, x, net, P, None)
    assert score.shape == ref.shape
    assert (score - ref).abs().max().item() < 1e-3
    assert abs(res[0].item() - loss.item()) < 1e-3
    assert res[1:].tolist() == pytest.approx(accs, abs=1e-6)
    assert torch.equal(eng.get_mask().cpu(), O.mask_closed_form(B, P, eng.SQ))
    for k, g in grads.items():  # relative L2: see DESIGN.md "ReLU-boundary flips"
        e = ((eng.G[k].cpu() - g).norm() / g.norm().clamp_min(1e-12)).item()
        assert e < 5e-2, (k, e)


def test_bf16_full_config4_shape_runs():
    """cfg4 shape (r34, 224^2) in throughput mode: finite loss/gradients, BN invariants (size-independent)."""
    B = 4
    eng = DPCEngine("resnet34", 224, 8, 5, 3, B, DEV, torch.bfloat16)
    eng.load_params(O.init_params_reference_style("resnet34", seed=0))
    x = torch.randn(B, 8, 3, 5, 224, 224, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    r0 = eng.train_step(x).cpu()
    for _ in range(3):
        r = eng.train_step(x).cpu()
    assert torch.isfinite(r).all() and torch.isfinite(eng.flat_g).all() and r[0] < r0[0]
    assert eng.R == B * 3 * 49


_GRAD_SCRIPT = """
import sys, torch
sys.path.insert(0, {root!r})
from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O
B = 16
eng = DPCEngine("resnet18", 128, 8, 5, 3, B, "cuda:0", torch.bfloat16)
eng.load_params(O.init_params_reference_style("resnet18", seed=1))
x = torch.randn(B, 8, 3, 5, 128, 128, device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(5))
eng.forward(x, train=True, dropout_masks=torch.ones(eng.n_steps, eng.M, eng.D, device="cuda:0"))
res = eng.loss_topk(True)
eng.backward()
torch.cuda.synchronize()
torch.save({{"g": eng.flat_g.cpu(), "offsets": eng.offsets, "res": res.cpu()}}, {out!r})
"""


def test_specialised_bf16_kernels_agree_with_generic_ones(tmp_path):
    """The bench geometry (r18, 128^2) at a batch large enough for every specialised bf16 kernel (loader/compute
    implicit GEMM, role-specialised patch kernel, staged-patch weight gradients) to be selected, against the same
    step on the generic kernels only (kernel selection is read from the environment once per process, hence two
    child processes).  Same arithmetic type on both sides, so the per-parameter gradients agree to the level of
    accumulation-order / ReLU-flip noise; a wrong kernel shows up as O(1)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, env in (("fast", {}), ("generic", {"DPC_IGEMM_WS": "0", "DPC_HALO_WS": "0", "DPC_WGRAD_PATCH": "0",
                                                "DPC_WGRAD_STEM": "0", "DPC_WGRAD_V1": "1"})):
        out = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, "-c", _GRAD_SCRIPT.format(root=root, out=out)], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(torch.load(out))
    a, b = outs
    assert abs(a["res"][0].item() - b["res"][0].item()) < 2e-2
    errs = {}
    for k, (o, n) in a["offsets"].items():
        ga, gb = a["g"][o:o + n], b["g"][o:o + n]
        errs[k] = ((ga - gb).norm() / gb.norm().clamp_min(1e-12)).item()
    print("specialised-vs-generic bf16 gradient rel-L2 (weights of conv1s):", {k: round(v, 4) for k, v in errs.items() if k.endswith("conv1.weight")})
    bad = {k: v for k, v in errs.items() if v > LIMIT}
    assert not bad, bad
