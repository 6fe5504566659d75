"""GPU tier of SURVEY.md section 8 f3: the LC classifier at the reference's widths on MI355X against fixtures produced by the
reference's own LC class (tests/golden/lc.npz): eval-mode output / context, a train-mode step (loss, accuracy, gradients,
running statistics), then the throughput mode and the drop-in module."""
import os

import numpy as np
import pytest
import torch

from dpc_amd.lc import LC, LCEngine
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def test_lc_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "lc.npz"), allow_pickle=False)
    eng = LCEngine("resnet18", 64, 8, 5, 2, DEV, torch.float32, num_class=101)
    assert eng.lib.kind == "hip"
    eng.load_params(O.make_lc_params_pcg("resnet18", 101))
    x = O.make_input_pcg(2, 8, 5, 64).to(DEV)
    target = torch.tensor([3, 77])
    out, ctx = eng.forward(x, target, train=False)
    assert (out.cpu() - torch.from_numpy(g["eval_output"])).abs().max().item() < TOL
    assert (ctx.cpu() - torch.from_numpy(g["eval_context"])).abs().max().item() < TOL
    ones_g = torch.ones(8, eng.M, eng.D, device=DEV)
    ones_f = torch.ones(2, eng.D, device=DEV)
    out, ctx = eng.forward(x, target, train=True, gru_masks=ones_g, fc_mask=ones_f)  # the golden step ran with p = 0 dropouts
    assert (out.cpu() - torch.from_numpy(g["train_output"])).abs().max().item() < TOL
    # BatchNorm1d over a batch of TWO clips: xhat = +-(x1 - x2) / sqrt((x1 - x2)^2 + 4 eps) amplifies the fp32 noise of x where
    # the two clips nearly agree (observed 1.3e-3 on one channel; the logits above stay within 1e-3)
    assert (ctx.cpu() - torch.from_numpy(g["train_context"])).abs().max().item() < 5 * TOL
    res = eng.result.cpu()
    assert abs(res[0].item() - g["train_loss_acc"][0]) < TOL and res[1].item() == pytest.approx(float(g["train_loss_acc"][1]))
    eng.backward()
    torch.cuda.synchronize()
    names = [str(n) for n in g["param_names"]]
    assert names == list(eng.G.keys())
    worst = 0.0
    for i, n in enumerate(names):
        gr = eng.G[n].cpu()
        assert gr.norm().item() == pytest.approx(float(g["grad_norm"][i]), rel=2e-2, abs=1e-6), n
        st = int(g["grad_substride::" + n])
        rs = torch.from_numpy(g["grad_sub::" + n])
        worst = max(worst, ((gr.flatten()[::st] - rs).norm() / rs.norm().clamp_min(1e-12)).item())
    print(f"LC gradients vs reference: worst rel-L2 {worst:.4f}")
    assert worst < 3e-2  # ReLU-boundary flips, as for DPC_RNN
    for k in ("backbone.bn1", "backbone.layer3.0.bn1", "backbone.layer4.1.bn2", "final_bn"):
        assert np.allclose(eng.BUF[k + ".running_mean"].cpu().numpy(), g["rm::" + k], rtol=1e-3, atol=1e-4), k
        assert np.allclose(eng.BUF[k + ".running_var"].cpu().numpy(), g["rv::" + k], rtol=1e-3, atol=1e-4), k
        assert int(eng.BUF[k + ".num_batches_tracked"]) == 1


def test_lc_bf16_training_and_module():
    B = 16
    eng = LCEngine("resnet18", 128, 8, 5, B, DEV, torch.bfloat16, num_class=101)
    m = LC(128, 8, 5, "resnet18", 0.5, 101, seed=0)
    eng.load_params({k: v.detach() for k, v in m.state_dict().items()})
    x = torch.randn(B, 8, 3, 5, 128, 128, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    target = torch.arange(B) % 101
    r0 = eng.train_step(x, target).clone()
    for _ in range(6):
        r = eng.train_step(x, target).clone()
    assert torch.isfinite(r).all() and torch.isfinite(eng.flat_g).all() and r[0].item() < r0[0].item()
    # drop-in module, eval mode (running statistics): LC(...).forward(block) -> (output, context)
    m = m.to(DEV).eval()
    out, ctx = m(x[:2])
    assert tuple(out.shape) == (2, 1, 101) and tuple(ctx.shape) == (2, 1, 256) and torch.isfinite(out).all()
