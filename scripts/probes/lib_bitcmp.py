"""Two builds of the kernel library on the same inputs, bit for bit: python scripts/probes/lib_bitcmp.py libA.so libB.so
conv_halo_ws variants (plain forward, input-gradient with fused reduction / gated addend + reduction, the stem), layer1 geometry of cfg2."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

libs = [L.Lib(os.path.abspath(p), "hip") for p in sys.argv[1:3]]
dev = torch.device("cuda:0")
BF = torch.bfloat16
torch.manual_seed(0)
N, T, H, W, Cc = 256, 5, 32, 32, 64


def run(lib, name):
    mode = 0 if name in ("fwd", "fwd+stats") else 1
    d = kc.conv_desc(BF, BF, mode, N, (T, H, W), (T, H, W), Cc, Cc, Cc, 9 * Cc, Cc, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    g = torch.Generator(device=dev).manual_seed(1)
    src = torch.randn(N, T, H, W, Cc, device=dev, generator=g).to(BF)
    wd = (torch.randn(Cc, 9 * Cc, device=dev, generator=g) * 0.05).to(BF)
    out = torch.zeros(N, T, H, W, Cc, device=dev, dtype=BF)
    addend = torch.randn(N, T, H, W, Cc, device=dev, generator=g).to(BF)
    raw = torch.randn(N, T, H, W, Cc, device=dev, generator=g).to(BF)
    nu = N * T * H * W * Cc // 8
    amask = torch.randint(0, 256, (nu,), device=dev, dtype=torch.uint8, generator=g)
    bmask = torch.randint(0, 256, (nu,), device=dev, dtype=torch.uint8, generator=g)
    mean, invstd = torch.randn(Cc, device=dev, generator=g) * 0.1, torch.rand(Cc, device=dev, generator=g) + 0.5
    rows = lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = torch.zeros(max(rows, 1), 2, Cc, device=dev)
    if name == "fwd":
        lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, lib.stream())
    elif name == "fwd+stats":
        lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), wd.data_ptr(), out.data_ptr(), None, stats.data_ptr(), lib.stream())
    else:
        ep = L.ConvEpilogue()
        if "add" in name:
            ep.addend, ep.addend_mask = addend.data_ptr(), amask.data_ptr()
        if "red" in name:
            ep.bn_raw, ep.bn_mask, ep.bn_mean, ep.bn_invstd, ep.stats = raw.data_ptr(), bmask.data_ptr(), mean.data_ptr(), invstd.data_ptr(), stats.data_ptr()
        lib.call("dpc_conv_igemm_ex", C.byref(d), src.data_ptr(), wd.data_ptr(), out.data_ptr(), C.byref(ep), lib.stream())
    torch.cuda.synchronize()
    if "red" in name:   # f64 reference of the fused BatchNorm-backward sums from the kernel's own (stored) output
        bits = ((bmask.view(-1, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(-1, Cc).double()
        dz = out.double().reshape(-1, Cc) * bits
        xh = (raw.double().reshape(-1, Cc) - mean.double()) * invstd.double()
        ref = torch.stack([dz.sum(0), (dz * xh).sum(0)])
        got = stats.double().sum(0)
        print(f"      {name}: fused sums vs f64 of the stored output: rel err {((got - ref).abs().max() / ref.abs().max()).item():.3g}")
    return out, stats, L.last_kernel(lib)


def stem(lib):
    n, h = 64, 64
    d = kc.conv_desc(BF, BF, 0, n, (T, h, h), (T, h, h), 16, 16, 64, 256, 64, (1, 4, 4), (1, 1, 1), (0, 2, 2))
    g = torch.Generator(device=dev).manual_seed(2)
    src = torch.randn(n, T, h, h, 16, device=dev, generator=g).to(BF)
    w = (torch.randn(64, 256, device=dev, generator=g) * 0.05).to(BF)
    out = torch.zeros(n, T, h, h, 64, device=dev, dtype=BF)
    rows = lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = torch.zeros(max(rows, 1), 2, 64, device=dev)
    lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), w.data_ptr(), out.data_ptr(), None, stats.data_ptr(), lib.stream())
    torch.cuda.synchronize()
    return out, stats, L.last_kernel(lib)


bad = 0
for rep in range(3):
    for name in ("fwd", "fwd+stats", "red", "add+red", "add", "stem"):
        res = [stem(lib) if name == "stem" else run(lib, name) for lib in libs]
        (o0, s0, k0), (o1, s1, k1) = res
        same_o = torch.equal(o0, o1)
        same_s = torch.equal(s0, s1)
        nd = (o0 != o1).sum().item()
        md = (o0.float() - o1.float()).abs().max().item()
        print(f"rep {rep} {name:10s} {k0:40s} out equal {same_o} ({nd} differ, max |d| {md:.3g})  stats equal {same_s}", flush=True)
        bad += (not same_o) + (not same_s)
        # self-consistency of each lib: run again, compare with itself
        o0b, s0b, _ = stem(libs[0]) if name == "stem" else run(libs[0], name)
        o1b, s1b, _ = stem(libs[1]) if name == "stem" else run(libs[1], name)
        if not (torch.equal(o0, o0b) and torch.equal(s0, s0b)):
            print("   lib A not repeatable", (o0 != o0b).sum().item())
        if not (torch.equal(o1, o1b) and torch.equal(s1, s1b)):
            print("   lib B not repeatable", (o1 != o1b).sum().item())
print("differences:", bad)
