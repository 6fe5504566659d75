"""GPU tier: the bf16 (throughput) mode's backward, block by block, against an oracle that rounds where the engine rounds.

The end-to-end bf16 gradients can only be held to the rounding noise of 17 / 33 stacked layers (30-46 % rel-L2 in layer1 / stem,
tests/golden/anchor_r18_128_b16.npz, anchor_r34_224_b4.npz) -- a dropped tap in one weight gradient (+33 %) would pass.  Here every BasicBlock type is
differentiated in isolation: the engine's own saved block input (bf16) and a random incoming gradient go through
``_Block.backward`` -- the specialised kernels the benchmark runs: role-specialised patch kernel with fused epilogues, plane and
loader/compute implicit GEMMs, staged-patch / transpose-read weight gradients, parity-class strided input-gradients -- and
through torch autograd of ``oracle.basic_block_rounded`` (f32 accumulation, bf16 storage points, straight-through rounding) on
the CPU.  What differs is the rounding of the backward intermediates only: every gradient is held to 2 % rel-L2, and a mutation
that drops one tap of a weight gradient is shown to fail the same check."""
import pytest
import torch

from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 0.02


def rel(a, b):
    return ((a.float().cpu() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


def nchw(t):  # engine activation [N,T,H,W,C] -> [N,C,T,H,W]
    return t.permute(0, 4, 1, 2, 3).contiguous()


def _filled(net, size, B):
    e = DPCEngine(net, size, 8, 5, 3, B, DEV, torch.bfloat16)
    e.load_params(O.init_params_reference_style(net, seed=0))
    x = O.make_input_pcg(B, 8, 5, size).to(DEV)
    e.forward(x, train=False)   # fills every block's saved tensors (x_in, raw, act1, masks, statistics)
    torch.cuda.synchronize()
    return e


@pytest.fixture(scope="module")
def eng():
    # B = 16: the batch at which every specialised bf16 kernel of cfg2 is selected (tests/test_plan.py pins the plan)
    return _filled("resnet18", 128, 16)


@pytest.fixture(scope="module")
def eng34():
    # the 224^2 family of BASELINE configs[3] / [4] (backbone/resnet_2d3d.py:278-284: resnet34 = [3, 4, 6, 3] blocks): 56^2 and 28^2
    # planes (wgrad_patch<64> / <32>), 14 x 14 x 3 and 7 x 7 x 2 volumes (padded-grid wgrad2, igemm_ws tiles that straddle clips)
    return _filled("resnet34", 224, 4)


@pytest.mark.parametrize("bi", [0, 1, 2, 3, 4, 5, 6, 7])
def test_block_backward_vs_rounding_oracle(eng, bi):
    _block_case(eng, bi)


# one block of every kind r34 / 224^2 has: layer1.0 / .1 (fold_prev) / .2, layer2.0 (strided + downsample) / .1 / .3, layer3.0 / .1 / .5,
# layer4.0 / .1 / .2 (no final ReLU)
@pytest.mark.parametrize("bi", [0, 1, 2, 3, 4, 6, 7, 8, 12, 13, 14, 15])
def test_block_backward_vs_rounding_oracle_r34_224(eng34, bi):
    _block_case(eng34, bi)


def _block_case(eng, bi):
    blk = eng.blocks[bi]
    pre = "backbone." + blk.site
    li, bj = int(blk.site[5]) - 1, int(blk.site.split(".")[1])
    g = torch.Generator().manual_seed(100 + bi)
    # an incoming gradient that is CORRELATED with the activations, as a loss gradient is (plus noise).  With a purely random one
    # the weight gradients are sums of uncorrelated products -- tiny against their own bf16 rounding noise (6 % on layer1.0.conv1
    # whose input, the pooled stem output, is non-negative): the check would measure the test's signal-to-noise, not the kernels.
    o = blk.out.float().cpu()
    dout = (0.05 * (o - o.mean()) / o.std() + 0.02 * torch.randn(tuple(o.shape), generator=g)).to(torch.bfloat16)
    # ---- oracle: autograd of the rounded block at the engine's saved input and the same weights
    p = {k: v.detach().cpu().clone().requires_grad_() for k, v in eng.PRM.items() if k.startswith(pre)}
    x = nchw(blk.x_in.float().cpu()).requires_grad_()
    out = O.basic_block_rounded(x, p, pre, li >= 2, 2 if (li > 0 and bj == 0) else 1, final_relu=blk.final_relu)
    assert rel(nchw(blk.out), out.detach()) < 5e-3        # same forward (the engine's stored output is the oracle's, bf16 rounding aside)
    out.backward(nchw(dout.float()))
    # ---- engine: the block's backward kernels on its own saved tensors
    for k in p:
        eng.G[k].zero_()
    if blk.prev is not None:
        blk.prev.c2.reduced_rows = 0
    dx = blk.backward(dout.to(DEV).clone(), need_dx=True)
    if blk.prev is not None:
        blk.prev.c2.reduced_rows = 0   # the fused reduction of the previous block's bn2 is not consumed here
        blk.prev.c2._coef_ready = False   # ... nor the coefficients the block prepared for it (plan_backward: fold_prev / reduce_first_prev)
    torch.cuda.synchronize()
    errs = {"dx": rel(nchw(dx), x.grad)}
    for k, v in p.items():
        errs[k[len(pre):]] = rel(eng.G[k], v.grad)
    worst = max(errs, key=errs.get)
    print(f"{pre} fold_c1={blk.fold_c1} gate={blk.gate} fold_prev={blk.fold_prev}: worst {worst} {errs[worst]:.4f}; dx {errs['dx']:.4f}")
    for k, e in errs.items():
        assert e < TOL, (pre + k, e)
    # ---- the check has teeth: one dropped tap of conv2's weight gradient is far outside the tolerance
    w = p[pre + "conv2.weight"].grad.clone()
    mut = eng.G[pre + "conv2.weight"].cpu().clone()
    mut[..., -1, -1] = 0
    assert rel(mut, w) > 5 * TOL


def test_fused_reduction_of_the_previous_block_is_what_the_standalone_pass_computes(eng):
    """layer1.1's first input-gradient carries the BatchNorm-backward partial sums of layer1.0's bn2 (dpc_conv_igemm_ex): finalised,
    they give the dgamma / dbeta the standalone reduction gives for the same gradient tensor"""
    import ctypes as C
    from dpc_amd import _lib as L
    blk, prev = eng.blocks[1], eng.blocks[0]
    if not blk.fold_prev:
        pytest.skip("the plan does not fold this reduction")
    g = torch.Generator().manual_seed(5)
    dout = (torch.randn(tuple(blk.out.shape), generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    dx = blk.backward(dout.clone(), need_dx=True)
    u = prev.c2
    # the block finalises the carried sums itself, before it forks its weight gradient (bn_prepare)
    assert u._coef_ready and u.reduced_rows == 0
    dgam, dbet = eng.G[u.bnname + ".weight"].clone(), eng.G[u.bnname + ".bias"].clone()
    u._coef_ready = False
    pr = C.c_int32(0)
    part = torch.empty_like(eng.stats)
    eng.call("dpc_bn_bwd_reduce", dx, None, u.mask, u.raw, L.BF16, u.rows, u.Co, u.mean, u.invstd, 1, part, C.byref(pr))
    dgam2, dbet2, coef2 = (torch.empty(n, device=DEV) for n in (u.Co, u.Co, 2 * u.Co))
    eng.call("dpc_bn_bwd_finalize", part, pr.value, u.Co, float(u.rows), dgam2, dbet2, coef2)
    torch.cuda.synchronize()
    u.reduced_rows = 0
    assert rel(dgam, dgam2.cpu()) < 1e-5 and rel(dbet, dbet2.cpu()) < 1e-5
