"""Kernel parity cases shared by the CPU tier (kernels executed by the host SIMT
simulator, tests/test_kernels_emu.py) and the GPU tier (the same C ABI of
libdpc_hip.so on an MI355X, tests/test_kernels_gpu.py).

Every case computes its expectation with plain torch CPU ops (f32, or f64 where
the quantity is a long reduction) -- the same ops the reference model runs
(backbone/resnet_2d3d.py, backbone/convrnn.py, dpc/model_3d.py, dpc/main.py) -- and
calls the kernel through dpc_amd._lib.  Tolerances: f32 mode 1e-4 relative to the
tensor's max-abs (the reference's own fp32 noise floor is ~1e-4);
bf16 mode 2^-7 relative (one bf16 rounding of the output).
"""
import ctypes as C

import torch
import torch.nn.functional as F

from dpc_amd import _lib as L


def cl(x):  # NCDHW -> channels-last contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous()


def tol(dtype):
    """max-abs error relative to the tensor's max-abs.  bf16: 2^-7 = two roundings of the largest element (the kernels accumulate in
    f32 and round once; an addend is one more) -- round 3 allowed 2^-6.  Every convolution case also proves that this bound has
    teeth: an expectation with ONE tap dropped must fail it (rejects_dropped_tap)."""
    return 1e-4 if dtype == torch.float32 else 1.0 / 128


def rejects_dropped_tap(got, want, want_without_tap, bound):
    """mutation check: the tolerance that accepts `got` against `want` must reject the same result against an expectation
    computed with one kernel tap zeroed (tolerances with no margin for a missing tap)"""
    e_ok, e_mut = relerr(got, want), relerr(got, want_without_tap)
    assert e_ok < bound <= e_mut * 0.25, f"tolerance {bound:.3g}: real error {e_ok:.3g}, error against the dropped-tap expectation only {e_mut:.3g}"


def q(x, dtype):  # quantise like the device tensor will be
    return x.to(dtype).float()


def bits_of(y, E):
    """byte-per-unit sign mask of a [rows, C] tensor (what dpc_bn_apply writes)"""
    b = (y.float().cpu().reshape(-1, E) > 0).to(torch.int32) * (1 << torch.arange(E, dtype=torch.int32))
    return b.sum(1)


def relerr(a, b):
    return (a.float().cpu() - b.float()).abs().max().item() / max(b.abs().max().item(), 1e-6)


class K:
    """binds a Lib + device"""

    def __init__(self, lib, device):
        self.lib, self.dev = lib, torch.device(device)

    def t(self, x, dtype=None):
        x = x.to(dtype) if dtype is not None else x
        return x.contiguous().to(self.dev)

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.dev)

    def call(self, name, *args):
        return self.lib.call(name, *args, self.lib.stream())

    def sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize()


def check_kernel(k: K, expect):
    """the kernel the last C-ABI call launched (dpc_last_kernel) is the variant the case is meant to cover.
    expect: None, a name prefix, or "prefix|substring" (both must match)"""
    if not expect:
        return
    got = L.last_kernel(k.lib)
    pre, _, sub = expect.partition("|")
    assert got.startswith(pre) and sub in got, f"case meant for {expect}, the library launched {got}"


def conv_desc(dtype_in, dtype_out, mode, N, R, S, Ci, src_ld, Co, ldw, ldo, k, s, p):
    return L.ConvDesc(L.dtype_code(dtype_in), L.dtype_code(dtype_out), mode, N, R[0], R[1], R[2], S[0], S[1], S[2],
                      Ci, src_ld, Co, ldw, ldo, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2])


# ------------------------------------------------------------------ conv forward (+ BN partial sums)
def case_conv_fwd(k: K, dtype, N, Ci, Co, T, H, W, ks, st, pd, seed=0, expect=None):
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(N, Ci, T, H, W, generator=g), dtype)
    w = q(torch.randn(Co, Ci, *ks, generator=g) * 0.1, dtype)
    y = F.conv3d(x, w, None, st, pd)
    To, Ho, Wo = y.shape[2:]
    taps = ks[0] * ks[1] * ks[2]
    d = conv_desc(dtype, dtype, 0, N, (To, Ho, Wo), (T, H, W), Ci, Ci, Co, taps * Ci, Co, ks, st, pd)
    src = k.t(cl(x), dtype)
    wp = k.t(w.permute(0, 2, 3, 4, 1).reshape(Co, taps * Ci), dtype)
    out = k.empty(N, To, Ho, Wo, Co, dtype=dtype)
    rows = k.lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = k.zeros(rows, 2, Co)
    k.call("dpc_conv_igemm", C.byref(d), src, wp, out, None, stats)
    check_kernel(k, expect)
    k.sync()
    assert relerr(out, cl(y)) < tol(dtype)
    if taps > 1:
        wm = w.clone()
        wm[:, :, ks[0] // 2, ks[1] // 2, ks[2] // 2] = 0   # the centre tap gone (the one tap every output position sees)
        rejects_dropped_tap(out, cl(y), cl(F.conv3d(x, wm, None, st, pd)), tol(dtype))
    o = out.float().cpu().reshape(-1, Co).double()
    s = stats.cpu().double()
    assert (s[:, 0].sum(0) - o.sum(0)).abs().max().item() < 1e-3 * max(1.0, o.abs().sum(0).max().item())
    assert (s[:, 1].sum(0) - (o * o).sum(0)).abs().max().item() < 1e-4 * (o * o).sum(0).max().item()


def case_conv_dgrad(k: K, dtype, N, Ci, Co, T, H, W, ks, st, pd, seed=1, expect=None, with_add=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, T, H, W, generator=g).requires_grad_()
    w = q(torch.randn(Co, Ci, *ks, generator=g) * 0.1, dtype)
    y = F.conv3d(x, w, None, st, pd)
    gy = q(torch.randn(y.shape, generator=g), dtype)
    gx = torch.autograd.grad(y, x, gy)[0]
    To, Ho, Wo = y.shape[2:]
    taps = ks[0] * ks[1] * ks[2]
    d = conv_desc(dtype, dtype, 1, N, (T, H, W), (To, Ho, Wo), Co, Co, Ci, taps * Co, Ci, ks, st, pd)
    add = q(torch.randn(N, T, H, W, Ci, generator=g), dtype)
    out = k.empty(N, T, H, W, Ci, dtype=dtype)
    wd = k.t(w.permute(1, 2, 3, 4, 0).reshape(Ci, taps * Co), dtype)
    k.call("dpc_conv_igemm", C.byref(d), k.t(cl(gy), dtype), wd, out, k.t(add, dtype) if with_add else None, None)
    check_kernel(k, expect)
    k.sync()
    want = cl(gx) + add if with_add else cl(gx)
    assert relerr(out, want) < tol(dtype)
    if taps > 1:
        wm = w.clone()
        wm[:, :, ks[0] // 2, ks[1] // 2, ks[2] // 2] = 0   # centre tap: a corner tap of a strided, padded conv may never touch the input
        gxm = cl(torch.autograd.grad(F.conv3d(x, wm, None, st, pd), x, gy)[0])
        rejects_dropped_tap(out, want, gxm + add if with_add else gxm, tol(dtype))


def case_conv_dgrad_alias(k: K, dtype, N, Ci, Co, T, H, W, ks, st, pd, seed=31, expect=None, gate=False):
    """the API's aliasing contract (include/dpc_hip.h): addend == out gives bit for bit what a separate addend buffer gives --
    through dpc_conv_igemm, and through dpc_conv_igemm_ex with the addend gated by a ReLU mask (the engine's in-place dx of a
    block's first conv, dpc_amd/engine.py _Block.backward)"""
    g = torch.Generator().manual_seed(seed)
    E = 8 if dtype == torch.bfloat16 else 4
    w = q(torch.randn(Co, Ci, *ks, generator=g) * 0.1, dtype)
    y_shape = F.conv3d(torch.zeros(N, Ci, T, H, W), w, None, st, pd).shape
    To, Ho, Wo = y_shape[2:]
    taps = ks[0] * ks[1] * ks[2]
    gy = k.t(cl(q(torch.randn(y_shape, generator=g), dtype)), dtype)
    d = conv_desc(dtype, dtype, 1, N, (T, H, W), (To, Ho, Wo), Co, Co, Ci, taps * Co, Ci, ks, st, pd)
    wd = k.t(w.permute(1, 2, 3, 4, 0).reshape(Ci, taps * Co), dtype)
    add = q(torch.randn(N, T, H, W, Ci, generator=g), dtype)
    mask = None
    if gate:
        mask = k.t(bits_of(torch.randn(N * T * H * W, Ci, generator=g), E).to(torch.uint8))

    def run(out, addend):
        if not gate:
            k.call("dpc_conv_igemm", C.byref(d), gy, wd, out, addend, None)
        else:
            ep = L.ConvEpilogue()
            ep.addend, ep.addend_mask = addend.data_ptr(), mask.data_ptr()
            k.call("dpc_conv_igemm_ex", C.byref(d), gy, wd, out, C.byref(ep))
        check_kernel(k, expect)
        k.sync()

    sep = k.empty(N, T, H, W, Ci, dtype=dtype)
    run(sep, k.t(add, dtype))
    inplace = k.t(add, dtype)
    run(inplace, inplace)
    assert torch.equal(sep.cpu(), inplace.cpu()), "addend aliased to out differs from the out-of-place result"


def case_conv_dgrad_ex(k: K, dtype, N, Ci, Co, T, H, W, ks, st, pd, gate=True, bnred=True, bn_relu=True, seed=7, expect=None, with_add=True):
    """dpc_conv_igemm_ex: input-gradient + (ReLU-gated residual addend) + (BatchNorm-backward partial sums of the unit whose
    output gradient it produces) in one launch == autograd's input-gradient, the addend masked by the sign mask, and the sums
    dpc_bn_bwd_reduce takes from the stored result (torch expectations, f64 sums)"""
    g = torch.Generator().manual_seed(seed)
    E = 8 if dtype == torch.bfloat16 else 4
    x = torch.randn(N, Ci, T, H, W, generator=g).requires_grad_()
    w = q(torch.randn(Co, Ci, *ks, generator=g) * 0.1, dtype)
    y = F.conv3d(x, w, None, st, pd)
    gy = q(torch.randn(y.shape, generator=g), dtype)
    gx = cl(torch.autograd.grad(y, x, gy)[0])                      # [N,T,H,W,Ci]
    To, Ho, Wo = y.shape[2:]
    taps = ks[0] * ks[1] * ks[2]
    d = conv_desc(dtype, dtype, 1, N, (T, H, W), (To, Ho, Wo), Co, Co, Ci, taps * Co, Ci, ks, st, pd)
    rows = N * T * H * W
    add = q(torch.randn(N, T, H, W, Ci, generator=g), dtype)
    act_out = q(torch.randn(rows, Ci, generator=g), dtype)          # the block output whose sign gates the addend
    raw = q(torch.randn(rows, Ci, generator=g) * 1.5 + 0.3, dtype)  # raw conv output of the unit being differentiated
    act_in = q(torch.randn(rows, Ci, generator=g), dtype)           # its activation (sign mask of the ReLU that follows it)
    mean, invstd = torch.randn(Ci, generator=g) * 0.2, torch.rand(Ci, generator=g) + 0.5
    amask, bmask = bits_of(act_out, E).to(torch.uint8), bits_of(act_in, E).to(torch.uint8)
    gate = gate and with_add
    want = gx + (add * (act_out.reshape(add.shape) > 0).float() if gate else add) if with_add else gx
    e = L.ConvEpilogue()
    keep = [k.t(add, dtype), k.t(amask), k.t(raw, dtype), k.t(bmask), k.t(mean), k.t(invstd)]
    nrows = k.lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = k.zeros(nrows, 2, Ci)
    e.addend, e.addend_mask = (keep[0].data_ptr() if with_add else None), (keep[1].data_ptr() if gate else None)
    if bnred:
        e.bn_raw, e.bn_mask, e.bn_mean, e.bn_invstd = keep[2].data_ptr(), (keep[3].data_ptr() if bn_relu else None), keep[4].data_ptr(), keep[5].data_ptr()
        e.stats = stats.data_ptr()
    out = k.empty(N, T, H, W, Ci, dtype=dtype)
    wd = k.t(w.permute(1, 2, 3, 4, 0).reshape(Ci, taps * Co), dtype)
    k.call("dpc_conv_igemm_ex", C.byref(d), k.t(cl(gy), dtype), wd, out, C.byref(e))
    check_kernel(k, expect)
    k.sync()
    assert relerr(out, want) < tol(dtype)
    if bnred:
        o = out.float().cpu().reshape(rows, Ci).double()            # sums are those of the STORED gradient
        dz = o * (act_in > 0).double() if bn_relu else o
        xh = (raw.double() - mean.double()) * invstd.double()
        sgot = stats.cpu().double()
        for got, exp in ((sgot[:, 0].sum(0), dz.sum(0)), (sgot[:, 1].sum(0), (dz * xh).sum(0))):
            assert (got - exp).abs().max().item() < 2e-4 * max(1.0, dz.abs().sum(0).max().item())


def case_conv_dgrad_inplace(k: K, dtype, N, Ci, Co, T, H, W, ks, st, pd, seed=5, expect=None):
    """input-gradient accumulated IN PLACE (addend == out): the engine adds a strided 1x1 downsample's gradient onto the
    main path's dx; positions no tap reaches must keep their value exactly"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, T, H, W, generator=g).requires_grad_()
    w = q(torch.randn(Co, Ci, *ks, generator=g) * 0.1, dtype)
    y = F.conv3d(x, w, None, st, pd)
    gy = q(torch.randn(y.shape, generator=g), dtype)
    gx = torch.autograd.grad(y, x, gy)[0]
    To, Ho, Wo = y.shape[2:]
    taps = ks[0] * ks[1] * ks[2]
    d = conv_desc(dtype, dtype, 1, N, (T, H, W), (To, Ho, Wo), Co, Co, Ci, taps * Co, Ci, ks, st, pd)
    base = q(torch.randn(N, T, H, W, Ci, generator=g), dtype)
    out = k.t(base.clone(), dtype)
    wd = k.t(w.permute(1, 2, 3, 4, 0).reshape(Ci, taps * Co), dtype)
    k.call("dpc_conv_igemm", C.byref(d), k.t(cl(gy), dtype), wd, out, out, None)
    check_kernel(k, expect)
    k.sync()
    assert relerr(out, cl(gx) + base) < tol(dtype)
    untouched = cl(gx) == 0
    assert torch.equal(out.cpu().float()[untouched], base[untouched])


def case_conv_wgrad(k: K, dtype, N, Ci, Co, T, H, W, ks, st, pd, seed=2, expect=None):
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(N, Ci, T, H, W, generator=g), dtype)
    w = (torch.randn(Co, Ci, *ks, generator=g) * 0.1).requires_grad_()
    y = F.conv3d(x, w, None, st, pd)
    gy = q(torch.randn(y.shape, generator=g), dtype)
    gw = torch.autograd.grad(y, w, gy)[0]
    To, Ho, Wo = y.shape[2:]
    taps = ks[0] * ks[1] * ks[2]
    d = conv_desc(dtype, torch.float32, 0, N, (To, Ho, Wo), (T, H, W), Ci, Ci, Co, taps * Ci, Co, ks, st, pd)
    ns = C.c_int32(0)
    k.call("dpc_conv_wgrad", C.byref(d), None, None, Co, None, C.byref(ns))
    part = k.zeros(ns.value, Co, taps * Ci)
    k.call("dpc_conv_wgrad", C.byref(d), k.t(cl(x), dtype), k.t(cl(gy), dtype), Co, part, C.byref(ns))
    check_kernel(k, expect)
    dw = k.zeros(*gw.shape)
    k.call("dpc_reduce_unpack", part, ns.value, dw, Co, taps, Ci, Ci * taps, 1, taps, 0)
    k.sync()
    assert relerr(dw, gw) < 1e-4  # f32 accumulation in both modes
    if taps > 1:
        gm = gw.clone()
        gm[:, :, ks[0] // 2, ks[1] // 2, ks[2] // 2] = 0
        rejects_dropped_tap(dw, gw, gm, 1e-4)


def case_gemm_nt(k: K, dtype, M, N, Kd, seed=3, expect=None):
    g = torch.Generator().manual_seed(seed)
    A = q(torch.randn(M, Kd, generator=g), dtype)
    B = q(torch.randn(N, Kd, generator=g), dtype)
    d = conv_desc(dtype, torch.float32, 0, M, (1, 1, 1), (1, 1, 1), Kd, Kd, N, Kd, N, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    out = k.empty(M, N)
    k.call("dpc_conv_igemm", C.byref(d), k.t(A, dtype), k.t(B, dtype), out, None, None)
    check_kernel(k, expect)
    k.sync()
    assert relerr(out, A.double() @ B.double().t()) < 1e-5


def case_reduce_unpack(k: K, ns, d0, d1, d2, permute, accumulate, expect=None, seed=5):
    """dpc_reduce_unpack: out[i0, i1, i2] (+)= sum over ns slabs of part[s][i0][i1][i2], written through arbitrary strides
    (permute = the conv-parameter layout [d0][d2][d1]); the few-slab / many-slab / transposing kernels"""
    g = torch.Generator().manual_seed(seed)
    part = torch.randn(ns, d0, d1, d2, generator=g)
    base = torch.randn(d0, d1, d2, generator=g)
    want = part.double().sum(0) + (base.double() if accumulate else 0)
    if permute:
        out = k.t(base.permute(0, 2, 1).contiguous())
        s0, s1, s2 = d1 * d2, 1, d1
    else:
        out = k.t(base.clone())
        s0, s1, s2 = d1 * d2, d2, 1
    k.call("dpc_reduce_unpack", k.t(part), ns, out, d0, d1, d2, s0, s1, s2, int(accumulate))
    check_kernel(k, expect)
    k.sync()
    got = out.cpu()
    got = got.permute(0, 2, 1) if permute else got
    assert relerr(got, want) < 1e-6


def case_gemm_nt_splitk(k: K, dtype, M, N, Kd, pad=0, seed=13, expect=None):
    """dpc_gemm_nt_splitk: f32 partial slabs of A @ B^T over K ranges, summed by dpc_reduce_unpack (d_pred = dS @ feature_inf);
    leading dimensions Kd + pad, the padding columns hold garbage that must not be read"""
    g = torch.Generator().manual_seed(seed)
    A = q(torch.randn(M, Kd, generator=g), dtype)
    B = q(torch.randn(N, Kd, generator=g), dtype)
    Ap = torch.full((M, Kd + pad), 77.0).to(dtype); Ap[:, :Kd] = A.to(dtype)
    Bp = torch.full((N, Kd + pad), -55.0).to(dtype); Bp[:, :Kd] = B.to(dtype)
    ns = C.c_int32(0)
    k.call("dpc_gemm_nt_splitk", L.dtype_code(dtype), M, N, Kd, None, Kd + pad, None, Kd + pad, None, C.byref(ns))
    part = k.zeros(ns.value, M, N)
    k.call("dpc_gemm_nt_splitk", L.dtype_code(dtype), M, N, Kd, k.t(Ap), Kd + pad, k.t(Bp), Kd + pad, part, C.byref(ns))
    check_kernel(k, expect)
    out = k.zeros(M, N)
    k.call("dpc_reduce_unpack", part, ns.value, out, M, 1, N, N, 0, 1, 0)
    k.sync()
    assert relerr(out, A.double() @ B.double().t()) < 1e-5
    return ns.value


def case_gemm_tn_splitk(k: K, M, N, Kd, pad=0, seed=17, expect="gemm_ws_kernel<true>"):
    """dpc_gemm_tn_splitk: out[m][n] = sum_k A[k][m] * B[n][k], A K-major with leading dimension M + pad (d_feature_inf = dS^T @ pred:
    A = dS [R][ld], B = pred^T [D][ld]); padding columns of both operands hold garbage that must not be read"""
    g = torch.Generator().manual_seed(seed)
    dtype = torch.bfloat16
    A = q(torch.randn(Kd, M, generator=g), dtype)
    B = q(torch.randn(N, Kd, generator=g), dtype)
    Ap = torch.full((Kd, M + pad), 77.0).to(dtype); Ap[:, :M] = A.to(dtype)
    Bp = torch.full((N, Kd + pad), -55.0).to(dtype); Bp[:, :Kd] = B.to(dtype)
    ns = C.c_int32(0)
    k.call("dpc_gemm_tn_splitk", L.dtype_code(dtype), M, N, Kd, None, M + pad, None, Kd + pad, None, C.byref(ns))
    part = k.zeros(ns.value, M, N)
    k.call("dpc_gemm_tn_splitk", L.dtype_code(dtype), M, N, Kd, k.t(Ap), M + pad, k.t(Bp), Kd + pad, part, C.byref(ns))
    check_kernel(k, expect)
    out = k.zeros(M, N)
    k.call("dpc_reduce_unpack", part, ns.value, out, M, 1, N, N, 0, 1, 0)
    k.sync()
    assert relerr(out, A.double().t() @ B.double().t()) < 1e-5
    return ns.value


def case_stem(k: K, dtype, BN, T, H, W, Co=64, seed=4, expect=(None, None)):
    """Conv3d(3,Co,(1,7,7),s(1,2,2),p(0,3,3)) (resnet_2d3d.py:211) through the space-to-depth path."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(BN, 3, T, H, W, generator=g)
    w = (torch.randn(Co, 3, 1, 7, 7, generator=g) * 0.1)
    wq = q(w, dtype).double().requires_grad_()  # f64 expectation: the f32 reference's own noise exceeds 1e-4 at full size
    y = F.conv3d(q(x, dtype).double(), wq, None, (1, 2, 2), (0, 3, 3))
    xs = k.empty(BN, T, H // 2, W // 2, 16, dtype=dtype)
    k.call("dpc_pack_input_s2d", k.t(x), xs, L.dtype_code(dtype), BN, T, H, W)
    wp = k.empty(Co, 16, 16, dtype=dtype)
    k.call("dpc_pack_stem_weight", k.t(w), wp, L.dtype_code(dtype), Co)
    d = conv_desc(dtype, dtype, 0, BN, (T, H // 2, W // 2), (T, H // 2, W // 2), 16, 16, Co, 256, Co, (1, 4, 4), (1, 1, 1), (0, 2, 2))
    out = k.empty(BN, T, H // 2, W // 2, Co, dtype=dtype)
    rows = k.lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = k.zeros(rows, 2, Co)
    k.call("dpc_conv_igemm", C.byref(d), xs, wp, out, None, stats)
    check_kernel(k, expect[0])
    k.sync()
    assert relerr(out, cl(y)) < tol(dtype)
    o = out.float().cpu().reshape(-1, Co).double()  # batch-norm partial sums are those of the STORED values
    st = stats.cpu().double()
    assert (st[:, 0].sum(0) - o.sum(0)).abs().max().item() < 1e-3 * max(1.0, o.abs().sum(0).max().item())
    assert (st[:, 1].sum(0) - (o * o).sum(0)).abs().max().item() < 1e-4 * (o * o).sum(0).max().item()
    gy = q(torch.randn(y.shape, generator=g), dtype)
    gw = torch.autograd.grad(y, wq, gy.double())[0]
    ns = C.c_int32(0)
    k.call("dpc_conv_wgrad", C.byref(d), None, None, Co, None, C.byref(ns))
    part = k.zeros(ns.value, Co, 256)
    k.call("dpc_conv_wgrad", C.byref(d), xs, k.t(cl(gy), dtype), Co, part, C.byref(ns))
    check_kernel(k, expect[1])
    dw = k.zeros(Co, 3, 1, 7, 7)
    k.call("dpc_unpack_stem_wgrad", part, ns.value, dw, Co)
    k.sync()
    assert relerr(dw, gw) < 1e-4


# ------------------------------------------------------------------ batch norm
def _bn_ref(x, gamma, beta, res, rs, rb, relu):
    """x [rows,C] f64: returns y, mean, invstd"""
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    y = (x - mean) * invstd * gamma + beta
    if res is not None:
        y = y + (res * rs + rb if rs is not None else res)
    if relu:
        y = F.relu(y)
    return y, mean, invstd


def case_bn_fwd_bwd(k: K, dtype, rows, Cc, relu, res_mode, seed=5):
    """res_mode: 0 none, 1 identity residual, 2 residual with its own scale/shift (downsample BN)"""
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(rows, Cc, generator=g) * 2 + 0.5, dtype)
    gamma = torch.rand(Cc, generator=g) + 0.5
    beta = torch.randn(Cc, generator=g) * 0.3
    res = q(torch.randn(rows, Cc, generator=g), dtype) if res_mode else None
    rs = torch.rand(Cc, generator=g) + 0.5 if res_mode == 2 else None
    rb = torch.randn(Cc, generator=g) if res_mode == 2 else None
    xd = x.double().requires_grad_()
    gd = gamma.double().requires_grad_()
    bd = beta.double().requires_grad_()
    y, mean, invstd = _bn_ref(xd, gd, bd, None if res is None else res.double(),
                              None if rs is None else rs.double(), None if rb is None else rb.double(), relu)
    # partial sums as the conv epilogue would deliver them (3 rows of partials)
    parts = torch.zeros(3, 2, Cc)
    for i, ch in enumerate(torch.chunk(x, 3, 0)):
        parts[i, 0] = ch.sum(0)
        parts[i, 1] = (ch * ch).sum(0)
    dm, di, dsc, dsh = (k.empty(Cc) for _ in range(4))
    k.call("dpc_bn_finalize", k.t(parts), 3, Cc, float(rows), k.t(gamma), k.t(beta), 1e-5,
           dm, di, dsc, dsh)
    k.sync()
    assert relerr(dm, mean.detach()) < 1e-5 and relerr(di, invstd.detach()) < 1e-4
    xk = k.t(x, dtype)
    yk = k.empty(rows, Cc, dtype=dtype)
    E = 4 if dtype == torch.float32 else 8
    mk = k.zeros(rows * Cc // E, dtype=torch.uint8)  # ReLU sign mask, one byte per 16-byte unit
    k.call("dpc_bn_apply", xk, yk, L.dtype_code(dtype), rows, Cc, dsc, dsh,
           None if res is None else k.t(res, dtype), None if rs is None else k.t(rs),
           None if rb is None else k.t(rb), int(relu), mk)
    k.sync()
    assert relerr(yk, y.detach()) < tol(dtype)
    bits = (yk.float().cpu().reshape(-1, E) > 0).to(torch.int32) * (1 << torch.arange(E, dtype=torch.int32))
    assert torch.equal(mk.cpu().to(torch.int32), bits.sum(1))
    # backward
    gy = q(torch.randn(rows, Cc, generator=g), dtype)
    y.backward(gy.double())
    yq = k.t(y.detach().float(), dtype)  # saved post-activation output (mask source)
    gyk = k.t(gy, dtype)
    prow = C.c_int32(0)
    k.call("dpc_bn_bwd_reduce", None, None, None, None, L.dtype_code(dtype), rows, Cc, None, None, int(relu), None, C.byref(prow))
    # the ReLU pattern comes either from the saved output y or from the byte mask of the kernel's own forward
    mq = k.t((bits_of(yq, E)).to(torch.uint8))
    for ysrc, msrc in ((yq, None), (None, mq)) if relu else ((None, None),):
        bp = k.zeros(prow.value, 2, Cc)
        k.call("dpc_bn_bwd_reduce", gyk, ysrc, msrc, xk, L.dtype_code(dtype), rows, Cc, dm, di,
               int(relu), bp, C.byref(prow))
        dgam, dbet, coef = k.empty(Cc), k.empty(Cc), k.empty(2, Cc)
        k.call("dpc_bn_bwd_finalize", bp, prow.value, Cc, float(rows), dgam, dbet, coef)
        dx = k.empty(rows, Cc, dtype=dtype)
        dz = k.empty(rows, Cc, dtype=dtype)
        k.call("dpc_bn_bwd_apply", gyk, ysrc, msrc, xk, L.dtype_code(dtype), rows, Cc, dm, di,
               k.t(gamma), coef, int(relu), dx, dz)
        k.sync()
        t = 2e-3 if dtype == torch.float32 else 3e-2  # relu-mask flips of y~0 elements are excluded by construction
        assert relerr(dgam, gd.grad) < t and relerr(dbet, bd.grad) < t
        assert relerr(dx, xd.grad) < t
        dz_ref = gy.double() * ((y.detach() > 0).double() if relu else 1.0)
        assert relerr(dz, dz_ref) < tol(dtype)


def case_bn_finalize(k: K, rows, Cc, misalign=0, seed=11):
    """dpc_bn_finalize / dpc_bn_bwd_finalize on a table of `rows` partial rows: f64 column sums (the unrolled four-row trips, the row tail,
    channel counts that are no multiple of four, a table that is not 16-byte aligned)"""
    g = torch.Generator().manual_seed(seed)
    parts = torch.randn(rows, 2, Cc, generator=g)
    parts[:, 1] = parts[:, 1].abs() * 3 + parts[:, 0] ** 2          # sum of squares >= (sum)^2 / n per row
    count = float(rows * 7)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    buf = k.zeros(rows * 2 * Cc + 4)
    pk = buf[misalign:misalign + rows * 2 * Cc].view(rows, 2, Cc)
    pk.copy_(parts)
    s1, s2 = parts[:, 0].double().sum(0), parts[:, 1].double().sum(0)
    m = s1 / count
    var = (s2 / count - m * m).clamp_min(0)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    dm, di, dsc, dsh = (k.empty(Cc) for _ in range(4))
    k.call("dpc_bn_finalize", pk, rows, Cc, count, k.t(gamma), k.t(beta), 1e-5, dm, di, dsc, dsh)
    dgam, dbet, coef = k.empty(Cc), k.empty(Cc), k.empty(2, Cc)
    k.call("dpc_bn_bwd_finalize", pk, rows, Cc, count, dgam, dbet, coef)
    k.sync()
    f = lambda t: t.float().cpu()  # noqa: E731
    assert torch.allclose(f(dm), m.float(), rtol=1e-6, atol=1e-7) and torch.allclose(f(di), inv.float(), rtol=1e-6)
    sc = gamma * inv.float()
    assert torch.allclose(f(dsc), sc, rtol=1e-6) and torch.allclose(f(dsh), beta - m.float() * sc, rtol=1e-5, atol=1e-6)
    assert torch.allclose(f(dbet), s1.float(), rtol=1e-6, atol=1e-6) and torch.allclose(f(dgam), s2.float(), rtol=1e-6)
    assert torch.allclose(f(coef[0]), (s1 / count).float(), rtol=1e-6, atol=1e-7) and torch.allclose(f(coef[1]), (s2 / count).float(), rtol=1e-6)


def case_stem_pool(k: K, dtype, NT, H, W, Cc, seed=6):
    """relu(bn(x)) -> MaxPool3d((1,3,3),(1,2,2),(0,1,1)) and its backward (resnet_2d3d.py:212-214)"""
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(NT, H, W, Cc, generator=g), dtype)
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.2
    a = (x * sc + sh).permute(0, 3, 1, 2).unsqueeze(2).requires_grad_()  # [NT,C,1,H,W]
    y = F.max_pool3d(F.relu(a), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    Ho, Wo = y.shape[3:]
    yk = k.empty(NT, Ho, Wo, Cc, dtype=dtype)
    am = torch.empty(NT, Ho, Wo, Cc, dtype=torch.uint8, device=k.dev)
    k.call("dpc_bn_relu_maxpool_fwd", k.t(x, dtype), L.dtype_code(dtype), NT, H, W, Cc, k.t(sc), k.t(sh),
           yk, am)
    k.sync()
    yref = y.detach().squeeze(2).permute(0, 2, 3, 1)
    assert relerr(yk, yref) < tol(dtype)
    gy = q(torch.randn(NT, Ho, Wo, Cc, generator=g), dtype)
    y.backward(gy.permute(0, 3, 1, 2).unsqueeze(2))
    dz = k.empty(NT, H, W, Cc, dtype=dtype)
    k.call("dpc_maxpool_bwd", k.t(gy, dtype), am, L.dtype_code(dtype), NT, H, W, Cc, dz)
    k.sync()
    dz_ref = a.grad.squeeze(2).permute(0, 2, 3, 1)
    assert relerr(dz, dz_ref) < tol(dtype)
    # fused form used by the engine: BN backward with the pooled gradient routed on the fly
    gamma = torch.rand(Cc, generator=g) + 0.5
    xd = x.double().requires_grad_()
    mean = xd.mean((0, 1, 2)); var = xd.var((0, 1, 2), unbiased=False); invstd = 1 / torch.sqrt(var + 1e-5)
    yb = (xd - mean) * invstd * gamma.double()
    yb.backward(dz_ref.double())
    rows = NT * H * W
    prow = C.c_int32(0)
    k.call("dpc_pool_bn_bwd_reduce", None, None, None, L.dtype_code(dtype), NT, H, W, Cc, None, None, None, C.byref(prow))
    bp = k.zeros(prow.value, 2, Cc)
    xk, gyk = k.t(x, dtype), k.t(gy, dtype)
    km, ki = k.t(mean.detach().float()), k.t(invstd.detach().float())
    k.call("dpc_pool_bn_bwd_reduce", gyk, am, xk, L.dtype_code(dtype), NT, H, W, Cc, km, ki, bp, C.byref(prow))
    dgam, dbet, coef = k.empty(Cc), k.empty(Cc), k.empty(2, Cc)
    k.call("dpc_bn_bwd_finalize", bp, prow.value, Cc, float(rows), dgam, dbet, coef)
    dx = k.empty(NT, H, W, Cc, dtype=dtype)
    k.call("dpc_pool_bn_bwd_apply", gyk, am, xk, L.dtype_code(dtype), NT, H, W, Cc, km, ki, k.t(gamma), coef, dx)
    k.sync()
    t = 2e-3 if dtype == torch.float32 else 3e-2
    assert relerr(dx, xd.grad) < t and relerr(dbet, dz_ref.double().sum((0, 1, 2))) < t
    # pooled-tensor form of the same partial sums (y at an argmax position is the pooled value): feed it the
    # scale/shift the forward used as gamma*invstd / beta-mean*scale, i.e. gamma'=sc, beta'=sh, xhat'=x
    prow2 = C.c_int32(0)
    rows_p = NT * Ho * Wo
    k.call("dpc_pooled_bn_bwd_reduce", None, None, None, L.dtype_code(dtype), rows_p, Cc, None, None, None, C.byref(prow2))
    bp2 = k.zeros(prow2.value, 2, Cc)
    k.call("dpc_pooled_bn_bwd_reduce", gyk, am, yk, L.dtype_code(dtype), rows_p, Cc, k.t(sc), k.t(sh), bp2, C.byref(prow2))
    k.sync()
    s_dz = bp2.cpu().double().sum(0)
    ref1 = dz_ref.double().sum((0, 1, 2))
    ref2 = (dz_ref.double() * x.double()).sum((0, 1, 2))   # with gamma'=sc, beta'=sh the recovered "xhat" is x itself
    assert (s_dz[0] - ref1).abs().max().item() < t * max(ref1.abs().max().item(), 1.0)
    assert (s_dz[1] - ref2).abs().max().item() < (t if dtype == torch.float32 else 6e-2) * max(ref2.abs().max().item(), 1.0)


def case_tpool_split(k: K, dtype, B, N, T, SQ, D, P, seed=7):
    g = torch.Generator().manual_seed(seed)
    x = q(torch.randn(B * N, T, SQ, D, generator=g), dtype).requires_grad_()
    m = x.mean(1).view(B, N, SQ, D)
    fr_ref = F.relu(m).permute(1, 0, 2, 3).reshape(N, B * SQ, D)
    fi_ref = m[:, N - P:]
    fr = k.empty(N, B * SQ, D, dtype=dtype)
    fi = k.empty(B, P, SQ, D, dtype=dtype)
    xk = k.t(x.detach(), dtype)
    k.call("dpc_tpool_split_fwd", xk, L.dtype_code(dtype), B, N, T, SQ, D, P, fr, fi)
    k.sync()
    assert relerr(fr, fr_ref.detach()) < tol(dtype) and relerr(fi, fi_ref.detach()) < tol(dtype)
    d_relu = torch.randn(N - P, B * SQ, D, generator=g)
    d_inf = torch.randn(B, P, SQ, D, generator=g)
    loss = (fr_ref[: N - P] * d_relu).sum() + (fi_ref * d_inf).sum()
    gx = torch.autograd.grad(loss, x)[0]
    dx = k.empty(B * N, T, SQ, D, dtype=dtype)
    k.call("dpc_tpool_split_bwd", xk, k.t(d_relu), k.t(d_inf), L.dtype_code(dtype), B, N, T, SQ, D, P, dx)
    k.sync()
    assert relerr(dx, gx) < tol(dtype)


# ------------------------------------------------------------------ ConvGRU cell pieces
def case_mask(k: K, B, P, SQ):
    from oracle import dpc_oracle as O
    m = torch.empty(B, P, SQ, B, P, SQ, dtype=torch.int8, device=k.dev)
    k.call("dpc_mask_gen", m, B, P, SQ)
    k.sync()
    assert torch.equal(m.cpu(), O.mask_closed_form(B, P, SQ))  # bit exact


def case_ce_topk(k: K, rows, cols, dtype_d, seed=10):
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(rows, cols, generator=g) * 3
    idx = torch.arange(0, rows, 3)
    s[idx, idx] += 6.0
    sd = s.double().requires_grad_()
    tgt = torch.arange(rows)
    loss = F.cross_entropy(sd, tgt)
    loss.backward()
    _, pred = s.topk(5, 1, True, True)
    correct = pred.t().eq(tgt.view(1, -1))
    accs = [correct[:kk].reshape(-1).float().sum().item() / rows for kk in (1, 3, 5)]
    ld_d = (cols + 7) // 8 * 8
    ws, res = k.empty(rows, 2), k.empty(4)
    ds = k.empty(rows, ld_d, dtype=dtype_d)
    k.call("dpc_ce_topk", k.t(s), rows, cols, cols, ws, res, ds, L.dtype_code(dtype_d), ld_d)
    k.sync()
    r = res.cpu()
    assert abs(r[0].item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    assert [round(v, 6) for v in r[1:].tolist()] == [round(v, 6) for v in accs]
    assert relerr(ds[:, :cols], sd.grad) < tol(dtype_d)
    if ld_d > cols:
        assert ds[:, cols:].float().abs().max().item() == 0


def case_adam(k: K, n, seed=11):
    from oracle import dpc_oracle as O
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g) * 0.01
    m = torch.randn(n, generator=g) * 0.01
    v = torch.rand(n, generator=g) * 1e-4
    pk, mk, vk = k.t(p.clone()), k.t(m.clone()), k.t(v.clone())
    step = 3
    k.call("dpc_adam", pk, k.t(gr), mk, vk, n, 1e-3, 0.9, 0.999, 1e-8, 1e-5,
           1 - 0.9 ** step, 1 - 0.999 ** step, 1.0)
    k.sync()
    O.adam_step(p, gr, m, v, step)
    assert (pk.cpu() - p).abs().max().item() < 1e-6 and relerr(mk, m) < 1e-5 and relerr(vk, v) < 1e-5


def case_transpose(k: K, rows, cols, seed=12):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(rows, cols, generator=g)
    ld = (rows + 7) // 8 * 8
    o = k.zeros(cols, ld, dtype=torch.bfloat16)
    k.call("dpc_transpose2d", k.t(a), 0, cols, o, 1, ld, rows, cols)
    k.sync()
    assert torch.equal(o[:, :rows].cpu(), a.t().bfloat16())


def case_transpose_x2(k: K, rows, cols, seed=21):
    """dpc_transpose2d_bf16x2: two bf16 matrices in one launch; padding columns of the outputs stay untouched (zero)"""
    g = torch.Generator().manual_seed(seed)
    a, b = torch.randn(rows, cols, generator=g).bfloat16(), torch.randn(rows, cols, generator=g).bfloat16()
    ld = (rows + 7) // 8 * 8
    oa, ob = k.zeros(cols, ld, dtype=torch.bfloat16), k.zeros(cols, ld, dtype=torch.bfloat16)
    k.call("dpc_transpose2d_bf16x2", k.t(a), k.t(b), cols, oa, ob, ld, rows, cols)
    k.sync()
    assert torch.equal(oa[:, :rows].cpu(), a.t()) and torch.equal(ob[:, :rows].cpu(), b.t())
    assert not oa[:, rows:].any() and not ob[:, rows:].any()
    oc = k.zeros(cols, ld, dtype=torch.bfloat16)
    k.call("dpc_transpose2d_bf16x2", k.t(b), None, cols, oc, None, ld, rows, cols)
    k.sync()
    assert torch.equal(oc[:, :rows].cpu(), b.t())


def case_pack3d_multi(k: K, dtype, shapes, seed=14):
    """dpc_pack3d_multi: every conv weight [Co][Ci][taps] f32 -> [Co][tap][Ci] and [Ci][tap][Co] in the compute dtype, one launch
    over a device-side table (the engine's per-step repack; tiled through LDS when the source is contiguous along the middle index)"""
    g = torch.Generator().manual_seed(seed)
    ents, keep, blk = [], [], 0
    for (Co, Ci, t) in shapes:
        w = k.t(torch.randn(Co, Ci, t, generator=g))
        wp, wd = k.zeros(Co, t, Ci, dtype=dtype), k.zeros(Ci, t, Co, dtype=dtype)
        for (dst, d0, d1, d2, s0, s1, s2) in ((wp, Co, t, Ci, Ci * t, 1, t), (wd, Ci, t, Co, t, 1, Ci * t)):
            ents.append(L.PackEntry(w.data_ptr(), dst.data_ptr(), d0, d1, d2, blk, s0, s1, s2))
            blk += max(1, min(1024, (d0 * d1 * d2 + 511) // 512))
        keep.append((w, wp, wd))
    tab = (L.PackEntry * len(ents))(*ents)
    tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).clone().to(k.dev)
    k.call("dpc_pack3d_multi", tab_dev, len(ents), blk, L.dtype_code(dtype))
    k.sync()
    for w, wp, wd in keep:
        assert torch.equal(wp.cpu(), w.cpu().permute(0, 2, 1).to(dtype))
        assert torch.equal(wd.cpu(), w.cpu().permute(1, 2, 0).to(dtype))


def case_copy2d_multi(k: K, seed=15):
    """dpc_copy2d_multi: a table of f32 [rows][cols] windows between different leading dimensions in one launch (the nine gate
    slices of the ConvGRU's parameter gradients); everything outside the windows stays untouched"""
    g = torch.Generator().manual_seed(seed)
    D = 24
    srcs = [k.t(torch.randn(3 * D, D, generator=g)), k.t(torch.randn(2 * D, D, generator=g)), k.t(torch.randn(3 * D, generator=g))]
    dsts = [k.t(torch.full((D, 2 * D), -7.0)) for _ in range(3)] + [k.t(torch.full((D,), -7.0)) for _ in range(3)]
    ents = []
    for i in range(3):
        ents.append((srcs[0][i * D:(i + 1) * D], D, dsts[i], 2 * D, D, D))                       # x half of gate i
        if i < 2:
            ents.append((srcs[1][i * D:(i + 1) * D], D, dsts[i][:, D:], 2 * D, D, D))            # h half
        ents.append((srcs[2][i * D:(i + 1) * D], D, dsts[3 + i], D, 1, D))                       # bias
    tab = (L.Copy2dEntry * len(ents))()
    blk = 0
    for i, (src, sld, dst, dld, r, c) in enumerate(ents):
        tab[i] = L.Copy2dEntry(src.data_ptr(), dst.data_ptr(), sld, dld, r, c, blk, 0)
        blk += 1 + (i % 2)
    tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).clone().to(k.dev)
    k.call("dpc_copy2d_multi", tab_dev, len(ents), blk)
    k.sync()
    for i in range(3):
        assert torch.equal(dsts[i][:, :D].cpu(), srcs[0][i * D:(i + 1) * D].cpu())
        if i < 2:
            assert torch.equal(dsts[i][:, D:].cpu(), srcs[1][i * D:(i + 1) * D].cpu())
        else:
            assert bool((dsts[i][:, D:] == -7.0).all())
        assert torch.equal(dsts[3 + i].cpu(), srcs[2][i * D:(i + 1) * D].cpu())


# ---------------------------------------------------------------- dropout masks (Philox4x32-10) / device-side Adam step
def philox4x32_10_np(ctr, key):
    """numpy Philox4x32-10 (Salmon et al., SC'11): ctr [n,4] uint32, key [2] uint32 -> [n,4] uint32.
    Pinned by the Random123 known-answer vectors in tests/test_kernels_emu.py::test_philox_known_answers."""
    import numpy as np
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, 1).astype(np.uint32)


def dropout_mask_np(n, p, seed, step):
    import numpy as np
    nb = (n + 3) // 4
    ctr = np.zeros((nb, 4), np.uint32)
    ctr[:, 0] = np.arange(nb, dtype=np.uint32)
    ctr[:, 1] = step
    r = philox4x32_10_np(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)[:n]
    thresh = int(p * 16777216.0 + 0.5)
    return ((r >> 8) >= thresh).astype(np.float32) / np.float32(1.0 - p)


def case_dropout_mask(k: K, n, p=0.1, seed=233, step=5):
    import numpy as np
    st = torch.tensor([step], dtype=torch.int32, device=k.dev)
    m = k.empty(n)
    k.call("dpc_dropout_mask", m, n, p, seed, st)
    k.sync()
    ref = dropout_mask_np(n, p, seed, step)
    assert np.array_equal(m.cpu().numpy(), ref)  # bit-exact
    if n >= 100000:
        assert abs(float((ref > 0).mean()) - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5
        other = dropout_mask_np(n, p, seed, step + 1)
        assert 0.15 < float((other != ref).mean()) < 0.21  # a new optimizer step draws new masks (2p(1-p) = 0.18)


def case_adam_dev(k: K, n, seed=13):
    from oracle import dpc_oracle as O
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(n, generator=g)
    m = torch.zeros(n)
    v = torch.zeros(n)
    pk, mk, vk = k.t(p.clone()), k.t(m.clone()), k.t(v.clone())
    st = torch.zeros(1, dtype=torch.int32, device=k.dev)
    bc = k.empty(2)
    for step in (1, 2, 3):
        gr = torch.randn(n, generator=g) * 0.01
        k.call("dpc_step_advance", st, bc, 0.9, 0.999)
        k.call("dpc_adam_dev", pk, k.t(gr), mk, vk, n, 1e-3, 0.9, 0.999, 1e-8, 1e-5, bc, 1.0)
        k.sync()
        O.adam_step(p, gr, m, v, step)
        assert int(st.item()) == step
        assert bc.cpu().tolist() == pytest_approx([1 - 0.9 ** step, 1 - 0.999 ** step])
    assert (pk.cpu() - p).abs().max().item() < 2e-6 and relerr(mk, m) < 1e-5 and relerr(vk, v) < 1e-5


def pytest_approx(v):
    import pytest
    return pytest.approx(v, rel=1e-6)


# ---------------------------------------------------------------- fused ConvGRU recurrence (csrc/gru_chain.hip)
def _chain_setup(k: K, dtype, B, SQ, D, P, n_agg, seed):
    g = torch.Generator().manual_seed(seed)
    ns, M = n_agg + P - 1, B * SQ
    w = {n: q(torch.randn(D, 2 * D, generator=g) * (1.0 / (2 * D) ** 0.5), dtype) for n in ("update", "reset", "out")}
    w["p0"] = q(torch.randn(D, D, generator=g) * (1.0 / D ** 0.5), dtype)
    w["p2"] = q(torch.randn(D, D, generator=g) * (1.0 / D ** 0.5), dtype)
    b = {n: torch.randn(D, generator=g) * 0.1 for n in ("update", "reset", "out", "p0", "p2")}
    x = q(torch.relu(torch.randn(n_agg, M, D, generator=g)), dtype)
    d_pred = torch.randn(B, P, SQ, D, generator=g) * 0.1
    dev = {}
    dev["packed"] = k.empty(16 * D * D, dtype=dtype)
    k.call("dpc_gru_pack", k.t(w["update"]), k.t(w["reset"]), k.t(w["out"]), k.t(w["p0"]), k.t(w["p2"]), D, L.dtype_code(dtype), dev["packed"])
    dev["bias"] = {n: k.t(v) for n, v in b.items()}
    X_all = k.zeros(ns, M, D, dtype=dtype)
    X_all[:n_agg] = k.t(x, dtype)
    dev.update(X_all=X_all, H_all=k.zeros(ns + 1, M, D, dtype=dtype), HR_all=k.empty(ns, M, D, dtype=dtype),
               U_all=k.empty(ns, M, D), R_all=k.empty(ns, M, D), O_all=k.empty(ns, M, D), P1_all=k.empty(P, M, D, dtype=dtype),
               pred=k.empty(B, P, SQ, D, dtype=dtype), d_pred=k.t(d_pred), G_all=k.empty(ns, M, 3 * D, dtype=dtype),
               dP1=k.empty(P, M, D, dtype=dtype), dP2=k.empty(P, M, D, dtype=dtype), d_x=k.empty(n_agg, M, D), ws=k.empty(2, M, D))
    d = L.GruChainDesc()
    d.dtype, d.M, d.D, d.SQ, d.P, d.n_agg, d.n_steps = L.dtype_code(dtype), M, D, SQ, P, n_agg, ns
    d.p_drop, d.seed = 0.1, 233
    d.packed = dev["packed"].data_ptr()
    for f, n in (("bias_u", "update"), ("bias_r", "reset"), ("bias_o", "out"), ("bias_1", "p0"), ("bias_2", "p2")):
        setattr(d, f, dev["bias"][n].data_ptr())
    for f in ("X_all", "H_all", "HR_all", "U_all", "R_all", "O_all", "P1_all", "pred", "d_pred", "G_all", "dP1", "dP2", "d_x", "ws"):
        setattr(d, f, dev[f].data_ptr())
    return d, dev, w, b, x, d_pred


def case_gru_chain(k: K, dtype, B, SQ, D, P, n_agg, seed=21):
    """dpc_gru_chain_fwd/_bwd with injected dropout masks against autograd over the reference's recurrence
    (dpc/model_3d.py:62-72 restated with 1x1 convolutions as matmuls on rows m = (b, s))."""
    d, dev, w, b, x, d_pred = _chain_setup(k, dtype, B, SQ, D, P, n_agg, seed)
    ns, M = n_agg + P - 1, B * SQ
    g = torch.Generator().manual_seed(seed + 1)
    masks = (torch.rand(ns, M, D, generator=g) > 0.1).float() / 0.9
    mdev = k.t(masks)
    d.drop_masks = mdev.data_ptr()
    k.call("dpc_gru_chain_fwd", C.byref(d))
    k.call("dpc_gru_chain_bwd", C.byref(d))
    k.sync()
    # ---- expectation (f64 autograd; operands quantised where the kernel stores them in `dtype`)
    W = {n: v.double().requires_grad_() for n, v in w.items()}
    Bs = {n: v.double().requires_grad_() for n, v in b.items()}
    xs = x.double().requires_grad_()
    qd = (lambda t: t) if dtype == torch.float32 else (lambda t: t + (t.detach().to(dtype).double() - t.detach()))  # straight-through rounding

    def cell(xi, h):
        c = torch.cat([xi, h], 1)
        u = torch.sigmoid(c @ W["update"].t() + Bs["update"])
        r = torch.sigmoid(c @ W["reset"].t() + Bs["reset"])
        o = torch.tanh(torch.cat([xi, qd(h * r)], 1) @ W["out"].t() + Bs["out"])
        return h * (1 - u) + o * u

    h = torch.zeros(M, D, dtype=torch.float64)
    step, preds, hs = 0, [], [h]
    for t in range(n_agg):
        h = qd(cell(xs[t], h) * masks[step].double())
        hs.append(h)
        step += 1
    for i in range(P):
        p1 = qd(torch.relu(h @ W["p0"].t() + Bs["p0"]))
        p2 = p1 @ W["p2"].t() + Bs["p2"]
        preds.append(qd(p2))
        if i < P - 1:
            h = qd(cell(qd(torch.relu(p2)), h) * masks[step].double())
            hs.append(h)
            step += 1
    pred = torch.stack(preds, 1).view(B, SQ, P, D).permute(0, 2, 1, 3)  # rows (b, s) -> [B][P][SQ][D]
    (pred * d_pred.double()).sum().backward()
    t_f, t_b = (2e-5, 2e-4) if dtype == torch.float32 else (3e-2, 6e-2)
    assert relerr(dev["pred"], pred.detach()) < t_f
    assert relerr(dev["H_all"][1:], torch.stack(hs[1:]).detach()) < t_f
    assert relerr(dev["d_x"], xs.grad) < t_b
    # the saved operands give the reference's weight gradients (the engine's batched GEMMs compute exactly these products)
    G = dev["G_all"].double().cpu().view(ns * M, 3 * D)
    Xa = dev["X_all"].double().cpu().view(ns * M, D)
    Ha = dev["H_all"][:ns].double().cpu().view(ns * M, D)
    HRa = dev["HR_all"].double().cpu().view(ns * M, D)
    gW = {"update": torch.cat([G[:, :D].t() @ Xa, G[:, :D].t() @ Ha], 1), "reset": torch.cat([G[:, D:2 * D].t() @ Xa, G[:, D:2 * D].t() @ Ha], 1),
          "out": torch.cat([G[:, 2 * D:].t() @ Xa, G[:, 2 * D:].t() @ HRa], 1)}
    Hp = dev["H_all"][n_agg:].double().cpu().view(P * M, D)
    gW["p0"] = dev["dP1"].double().cpu().view(P * M, D).t() @ Hp
    gW["p2"] = dev["dP2"].double().cpu().view(P * M, D).t() @ dev["P1_all"].double().cpu().view(P * M, D)
    for n in gW:
        assert relerr(gW[n], W[n].grad) < t_b, n
    gB = {"update": G[:, :D].sum(0), "reset": G[:, D:2 * D].sum(0), "out": G[:, 2 * D:].sum(0),
          "p0": dev["dP1"].double().cpu().view(P * M, D).sum(0), "p2": dev["dP2"].double().cpu().view(P * M, D).sum(0)}
    for n in gB:
        assert relerr(gB[n], Bs[n].grad) < t_b, n


def case_gru_chain_philox(k: K, dtype, B, SQ, D, P, n_agg, seed=22):
    """train mode: masks generated in the kernel (Philox keyed on seed + device-side step) == the masks dpc_dropout_mask
    writes for the same (seed, step), injected explicitly: bit-identical states and gradients; eval mode = no mask."""
    ns, M = n_agg + P - 1, B * SQ
    step = torch.tensor([3], dtype=torch.int32, device=k.dev)
    d1, dev1, *_ = _chain_setup(k, dtype, B, SQ, D, P, n_agg, seed)
    d1.step_dev = step.data_ptr()
    k.call("dpc_gru_chain_fwd", C.byref(d1))
    k.call("dpc_gru_chain_bwd", C.byref(d1))
    d2, dev2, *_ = _chain_setup(k, dtype, B, SQ, D, P, n_agg, seed)
    masks = k.empty(ns, M, D)
    k.call("dpc_dropout_mask", masks, masks.numel(), 0.1, 233, step)
    d2.drop_masks = masks.data_ptr()
    k.call("dpc_gru_chain_fwd", C.byref(d2))
    k.call("dpc_gru_chain_bwd", C.byref(d2))
    k.sync()
    for f in ("H_all", "pred", "d_x", "G_all", "dP1"):
        assert torch.equal(dev1[f], dev2[f]), f
    assert 0.85 < (masks > 0).float().mean().item() < 0.95
    d3, dev3, *_ = _chain_setup(k, dtype, B, SQ, D, P, n_agg, seed)
    k.call("dpc_gru_chain_fwd", C.byref(d3))  # eval: neither masks nor a step counter
    k.sync()
    assert not torch.equal(dev3["H_all"], dev1["H_all"])


def case_gru_chain_golden(k: K, ops):
    """The reference's own ConvGRUCell(8, 8, 1) fixture (tests/golden/ops.npz "gru::*", forward + autograd of
    backbone/convrnn.py:24-34) through the fused recurrence kernels: one aggregation step from the fixture's non-zero
    h, the 8 channels zero-padded to the kernel's 32-channel granularity (padded channels stay exactly 0), and an
    identity predictor (W1 = W2 = I, b1 = +4, b2 = -4: relu is transparent for |h| < 1) so that d_pred IS d/dh'."""
    f32 = torch.float32
    x, h = torch.from_numpy(ops["gru::x"]), torch.from_numpy(ops["gru::h"])      # [3, 8, 3, 3]
    Bn, Cg, ls = x.shape[0], x.shape[1], x.shape[2]
    D, SQ, M = 32, ls * ls, Bn * ls * ls
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(M, Cg)
    pad = lambda t: torch.cat([t, torch.zeros(t.shape[0], D - Cg)], 1)
    d, dev, *_ = _chain_setup(k, f32, Bn, SQ, D, 1, 1, seed=1)
    w = {}
    for gate in ("update", "reset", "out"):
        wg = torch.from_numpy(ops[f"gru::w::{gate}_gate.weight"]).view(Cg, 2 * Cg)
        full = torch.zeros(D, 2 * D)
        full[:Cg, :Cg], full[:Cg, D:D + Cg] = wg[:, :Cg], wg[:, Cg:]
        w[gate] = full
    eye = torch.eye(D)
    k.call("dpc_gru_pack", k.t(w["update"]), k.t(w["reset"]), k.t(w["out"]), k.t(eye), k.t(eye), D, L.F32, dev["packed"])
    for f, n in (("bias_u", "update"), ("bias_r", "reset"), ("bias_o", "out")):
        bt = k.t(torch.cat([torch.from_numpy(ops[f"gru::w::{n}_gate.bias"]), torch.zeros(D - Cg)]))
        dev["bias"][f] = bt
        setattr(d, f, bt.data_ptr())
    b1, b2 = k.t(torch.full((D,), 4.0)), k.t(torch.full((D,), -4.0))
    d.bias_1, d.bias_2 = b1.data_ptr(), b2.data_ptr()
    dev["X_all"][0] = k.t(pad(rows(x)))
    dev["H_all"][0] = k.t(pad(rows(h)))
    gh = torch.from_numpy(ops["gru::gh"])
    dev["d_pred"].copy_(k.t(pad(rows(gh)).view(Bn, SQ, 1, D).permute(0, 2, 1, 3).contiguous()))
    k.call("dpc_gru_chain_fwd", C.byref(d))
    k.call("dpc_gru_chain_bwd", C.byref(d))
    k.sync()
    hn = dev["H_all"][1].cpu()
    assert (hn[:, :Cg] - rows(torch.from_numpy(ops["gru::hn"]))).abs().max().item() < 1e-5
    assert hn[:, Cg:].abs().max().item() == 0
    assert (dev["pred"].cpu().view(Bn, SQ, D).reshape(M, D) - hn).abs().max().item() < 1e-5
    assert (dev["d_x"][0].cpu()[:, :Cg] - rows(torch.from_numpy(ops["gru::gx"]))).abs().max().item() < 1e-5
    G = dev["G_all"][0].cpu()
    Xr, Hr, HRr = pad(rows(x)), pad(rows(h)), dev["HR_all"][0].cpu()
    for gi, gate in enumerate(("update", "reset", "out")):
        Gg = G[:, gi * D:gi * D + Cg]
        gw = torch.cat([Gg.t() @ Xr[:, :Cg], Gg.t() @ (HRr if gate == "out" else Hr)[:, :Cg]], 1)
        ref = torch.from_numpy(ops[f"gru::gw::{gate}_gate.weight"]).view(Cg, 2 * Cg)
        assert (gw - ref).abs().max().item() < 1e-4, gate
        assert (Gg.sum(0) - torch.from_numpy(ops[f"gru::gw::{gate}_gate.bias"])).abs().max().item() < 1e-4, gate


# ---------------------------------------------------------------- fused score + CE/top-k + backward (csrc/score_fused.hip)
def case_score_fused(k: K, R, D, seed=31, check_score=True):
    """dpc_score_fwd / dpc_ce_finalize / dpc_score_bwd (bf16 operands, nothing [R][R] in memory) against the reference's
    formulation: score = pred @ finf^T (dpc/model_3d.py:83), CrossEntropyLoss with target = arange, calc_topk_accuracy,
    and autograd's d/dpred, d/dfinf -- computed in f64 from the same bf16-rounded operands."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    pred = (torch.randn(R, D, generator=g) * (2.0 / D ** 0.5)).to(bf)
    finf = (torch.randn(R, D, generator=g) * (2.0 / D ** 0.5)).to(bf)
    idx = torch.arange(0, R, 3)
    finf[idx] = (finf[idx].float() + 1.5 * pred[idx].float()).to(bf)  # a third of the targets rank high
    pd, fd = pred.double().requires_grad_(), finf.double().requires_grad_()
    S = pd @ fd.t()
    tgt = torch.arange(R)
    loss = F.cross_entropy(S, tgt)
    loss.backward()
    rank = (S.detach() > S.detach().diagonal()[:, None]).sum(1)
    accs = [(rank < kk).double().mean().item() for kk in (1, 3, 5)]
    ld = (R + 7) // 8 * 8
    predT, finfT = k.zeros(D, ld, dtype=bf), k.zeros(D, ld, dtype=bf)
    predT[:, :R] = k.t(pred.t().contiguous())
    finfT[:, :R] = k.t(finf.t().contiguous())
    pk, fk = k.t(pred), k.t(finf)
    nf, nb = C.c_int64(0), C.c_int64(0)
    ns = k.lib.call("dpc_score_ws_floats", R, D, C.byref(nf), C.byref(nb))
    assert ns >= 1
    ws = k.empty(max(nf.value, nb.value))
    diag, lse2, row_ws, res = k.empty(R), k.empty(R), k.empty(R, 2), k.empty(4)
    score = k.empty(R, R) if check_score else None
    k.call("dpc_score_fwd", pk, fk, R, D, diag, lse2, row_ws, score, ws)
    k.call("dpc_ce_finalize", row_ws, R, res)
    k.sync()
    Sd = S.detach()
    if check_score:
        assert relerr(score, Sd) < 1e-5
    assert (diag.cpu().double() - Sd.diagonal()).abs().max().item() < 1e-4 * max(1.0, Sd.abs().max().item())
    lse = torch.logsumexp(Sd, 1)
    assert (row_ws[:, 0].cpu().double() - (lse - Sd.diagonal())).abs().max().item() < 2e-4
    # a logit within f32 rounding of the target may fall on either side: ranks may differ on a few rows by one
    dr = (row_ws[:, 1].cpu().double() - rank.double()).abs()
    assert dr.max().item() <= 1 and (dr > 0).double().mean().item() < 5e-3
    r = res.cpu()
    assert abs(r[0].item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    for got, exp in zip(r[1:].tolist(), accs):
        assert abs(got - exp) <= 3.0 / R + 1e-6
    outs = {}
    for name, own, oth, othT, by_owner in (("d_pred", pk, fk, finfT, 1), ("d_finf", fk, pk, predT, 0)):
        n = k.lib.call("dpc_score_bwd", own, oth, othT, ld, R, D, lse2, by_owner, ws, k.lib.stream())
        assert n == ns
        out = k.empty(R, D)
        k.call("dpc_reduce_unpack", ws, n, out, R, 1, D, D, 0, 1, 0)
        outs[name] = out
    k.sync()
    # dS is rounded to bf16 before the second product: 2^-8 relative per term, averaged over R terms
    assert relerr(outs["d_pred"], pd.grad) < 1.5e-2
    assert relerr(outs["d_finf"], fd.grad) < 1.5e-2


def case_stem_wgrad_fused(k: K, BN, T, H, W, Co=64, seed=8):
    """dpc_stem_wgrad_fused (bf16): the stem's weight gradient computed from the gradient at the POOLED output with the
    max-pool routing and the BatchNorm backward inside the weight-gradient kernel, against (a) the two-kernel form
    dpc_pool_bn_bwd_apply -> dpc_conv_wgrad of the same library (bit-identical: same bf16 dz operand, same chunking) and
    (b) f64 autograd through conv -> batch-stat BN -> ReLU -> MaxPool3d (resnet_2d3d.py:211-214)."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(BN, 3, T, H, W, generator=g)
    w = torch.randn(Co, 3, 1, 7, 7, generator=g) * 0.1
    gamma, beta = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2
    Hs, Ws = H // 2, W // 2
    dc = L.BF16
    xs = k.empty(BN, T, Hs, Ws, 16, dtype=bf)
    k.call("dpc_pack_input_s2d", k.t(x), xs, dc, BN, T, H, W)
    wp = k.empty(Co, 16, 16, dtype=bf)
    k.call("dpc_pack_stem_weight", k.t(w), wp, dc, Co)
    d = conv_desc(bf, bf, 0, BN, (T, Hs, Ws), (T, Hs, Ws), 16, 16, Co, 256, Co, (1, 4, 4), (1, 1, 1), (0, 2, 2))
    dw_ = conv_desc(bf, torch.float32, 0, BN, (T, Hs, Ws), (T, Hs, Ws), 16, 16, Co, 256, Co, (1, 4, 4), (1, 1, 1), (0, 2, 2))
    raw = k.empty(BN, T, Hs, Ws, Co, dtype=bf)
    rows = k.lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = k.zeros(rows, 2, Co)
    k.call("dpc_conv_igemm", C.byref(d), xs, wp, raw, None, stats)
    M = BN * T * Hs * Ws
    mean, invstd, scale, shift = (k.empty(Co) for _ in range(4))
    k.call("dpc_bn_finalize", stats, rows, Co, float(M), k.t(gamma), k.t(beta), 1e-5, mean, invstd, scale, shift)
    Ho, Wo = (Hs - 1) // 2 + 1, (Ws - 1) // 2 + 1
    pooled = k.empty(BN * T, Ho, Wo, Co, dtype=bf)
    am = torch.empty(BN * T, Ho, Wo, Co, dtype=torch.uint8, device=k.dev)
    k.call("dpc_bn_relu_maxpool_fwd", raw, dc, BN * T, Hs, Ws, Co, scale, shift, pooled, am)
    gy = q(torch.randn(BN * T, Ho, Wo, Co, generator=g), bf)
    gyk = k.t(gy, bf)
    prow = C.c_int32(0)
    k.call("dpc_pooled_bn_bwd_reduce", None, None, None, dc, BN * T * Ho * Wo, Co, None, None, None, C.byref(prow))
    bp = k.zeros(prow.value, 2, Co)
    k.call("dpc_pooled_bn_bwd_reduce", gyk, am, pooled, dc, BN * T * Ho * Wo, Co, k.t(gamma), k.t(beta), bp, C.byref(prow))
    dgam, dbet, coef = k.empty(Co), k.empty(Co), k.empty(2, Co)
    k.call("dpc_bn_bwd_finalize", bp, prow.value, Co, float(M), dgam, dbet, coef)
    # (a) two kernels
    dz = k.empty(BN, T, Hs, Ws, Co, dtype=bf)
    k.call("dpc_pool_bn_bwd_apply", gyk, am, raw, dc, BN * T, Hs, Ws, Co, mean, invstd, k.t(gamma), coef, dz)
    ns = C.c_int32(0)
    k.call("dpc_conv_wgrad", C.byref(dw_), None, None, Co, None, C.byref(ns))
    part = k.zeros(ns.value, Co, 256)
    k.call("dpc_conv_wgrad", C.byref(dw_), xs, dz, Co, part, C.byref(ns))
    dw_a = k.zeros(Co, 3, 1, 7, 7)
    k.call("dpc_unpack_stem_wgrad", part, ns.value, dw_a, Co)
    # (b) fused
    ns2 = C.c_int32(0)
    rc = k.lib.call("dpc_stem_wgrad_fused", C.byref(dw_), None, None, None, None, None, None, None, None, None, C.byref(ns2), k.lib.stream())
    assert rc == 0 and ns2.value == ns.value
    part2 = k.zeros(ns2.value, Co, 256)
    k.call("dpc_stem_wgrad_fused", C.byref(dw_), xs, raw, gyk, am, mean, invstd, k.t(gamma), coef, part2, C.byref(ns2))
    dw_b = k.zeros(Co, 3, 1, 7, 7)
    k.call("dpc_unpack_stem_wgrad", part2, ns2.value, dw_b, Co)
    k.sync()
    assert torch.equal(dw_a, dw_b)
    # (c) autograd (bf16 quantisation of operands / dz: loose bound; the exactness claim is (a) == (b))
    xq, wq = q(x, bf).double(), q(w, bf).double().requires_grad_()
    y = F.conv3d(xq, wq, None, (1, 2, 2), (0, 3, 3))
    mu = y.mean((0, 2, 3, 4), keepdim=True); var = y.var((0, 2, 3, 4), unbiased=False, keepdim=True)
    z = (y - mu) / torch.sqrt(var + 1e-5) * gamma.double().view(1, -1, 1, 1, 1) + beta.double().view(1, -1, 1, 1, 1)
    pz = F.max_pool3d(F.relu(z), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    gyr = gy.double().view(BN, T, Ho, Wo, Co).permute(0, 4, 1, 2, 3)
    gw = torch.autograd.grad(pz, wq, gyr)[0]
    assert relerr(dw_b, gw) < 0.15  # sanity: bf16 raw / dz and argmax near-ties at a tiny batch (observed 0.09, identical for (a))


# ------------------------------------------------------------------ bf16 logits of the train step (round 6)
def case_ce_topk_bf16(k: K, rows, cols, seed=10):
    """dpc_ce_topk_bf16: loss, top-1/3/5 and d(loss)/d(score) of bf16-ROUNDED logits (the expectation is computed on the rounded
    values: what the kernel is given); exact ties at the target's value rank behind it"""
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(rows, cols, generator=g) * 3
    idx = torch.arange(0, rows, 3)
    s[idx, idx] += 6.0
    sq = s.to(torch.bfloat16)
    sd = sq.double().requires_grad_()
    tgt = torch.arange(rows)
    loss = F.cross_entropy(sd, tgt)
    loss.backward()
    rank = (sq.float() > sq.float().diagonal().view(-1, 1)[:, :1].expand(rows, cols)).sum(1)
    accs = [(rank < kk).float().mean().item() for kk in (1, 3, 5)]
    ld_d = (cols + 7) // 8 * 8
    ws, res = k.empty(rows, 2), k.empty(4)
    ds = k.empty(rows, ld_d, dtype=torch.bfloat16)
    k.call("dpc_ce_topk_bf16", k.t(sq, torch.bfloat16), rows, cols, cols, ws, res, ds, ld_d)
    k.sync()
    r = res.cpu()
    assert abs(r[0].item() - loss.item()) < 2e-5 * max(1.0, abs(loss.item()))
    assert [round(v, 6) for v in r[1:].tolist()] == [round(v, 6) for v in accs]
    assert relerr(ds[:, :cols], sd.grad) < tol(torch.bfloat16)


def case_gemm_nt_bf16out(k: K, M, N, Kd, seed=3, expect="score_gemm2_kernel<KS,OUT16>"):
    """the materialised score in the compute dtype: A @ B^T rounded once to bf16 (whole 128-byte rows: N a multiple of 64)"""
    g = torch.Generator().manual_seed(seed)
    dtype = torch.bfloat16
    A = q(torch.randn(M, Kd, generator=g), dtype)
    B = q(torch.randn(N, Kd, generator=g), dtype)
    d = conv_desc(dtype, dtype, 0, M, (1, 1, 1), (1, 1, 1), Kd, Kd, N, Kd, N, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    out = k.empty(M, N, dtype=dtype)
    k.call("dpc_conv_igemm", C.byref(d), k.t(A, dtype), k.t(B, dtype), out, None, None)
    check_kernel(k, expect)
    k.sync()
    want = (A.double() @ B.double().t())
    assert relerr(out, want) < 2.0 ** -8          # one rounding of an f32 accumulator
    assert torch.equal(out.cpu(), want.float().to(dtype)) or (out.cpu().float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()
