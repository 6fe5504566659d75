"""Does the 256 MB Infinity Cache pay for a batch-chunked layer1 backward?  A layer1 unit's backward streams 671 MB tensors (cfg2,
batch 128): bn_bwd_apply writes dz, the input-gradient and the weight gradient read it back from HBM.  In k chunks of 671 / k MB the
consumers could find dz (and the apply its freshly written dy) on the die.  Timed: the same three launches over the whole batch and
over k = 2, 4, 8 chunks (descriptor N / k, offset pointers), one stream, events around the whole sequence."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

lib = L.load_hip()
dev = torch.device("cuda:0")
BF = torch.bfloat16
N, T, H, W, Cc = 1024, 5, 32, 32, 64
if len(sys.argv) > 1 and sys.argv[1] == "layer2":
    N, T, H, W, Cc = 1024, 5, 16, 16, 128
per = T * H * W * Cc            # elements per frame group
dy = torch.randn(N, T, H, W, Cc, device=dev).to(BF)
raw = torch.randn(N, T, H, W, Cc, device=dev).to(BF)
x_in = torch.randn(N, T, H, W, Cc, device=dev).to(BF)
dz = torch.empty(N, T, H, W, Cc, device=dev, dtype=BF)
dx = torch.empty(N, T, H, W, Cc, device=dev, dtype=BF)
mask = torch.randint(0, 256, (N * per // 8,), device=dev, dtype=torch.uint8)
wd = (torch.randn(Cc, 9 * Cc, device=dev) * 0.05).to(BF)
mean, invstd, gamma = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev), torch.ones(Cc, device=dev)
coef = torch.zeros(2 * Cc, device=dev)
dc = L.dtype_code(BF)
esz = 2


def descs(n):
    dd = kc.conv_desc(BF, BF, 1, n, (T, H, W), (T, H, W), Cc, Cc, Cc, 9 * Cc, Cc, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    dw = kc.conv_desc(BF, torch.float32, 0, n, (T, H, W), (T, H, W), Cc, Cc, Cc, 9 * Cc, Cc, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    return dd, dw


def part_for(dw):
    ns = C.c_int32(0)
    lib.call("dpc_conv_wgrad", C.byref(dw), None, None, Cc, None, C.byref(ns), lib.stream())
    return ns.value


ns_full = part_for(descs(N)[1])
part = torch.empty(max(ns_full, 1) * Cc * 9 * Cc, device=dev)
print('last kernel', L.last_kernel(lib))


def seq(k, with_wgrad, with_apply=True, with_dgrad=True):
    n = N // k
    dd, dw = descs(n)
    ns = C.c_int32(0)
    for c in range(k):
        o = c * n * per
        if with_apply:
            lib.call("dpc_bn_bwd_apply", dy.data_ptr() + o * esz, None, mask.data_ptr() + o // 8, raw.data_ptr() + o * esz, dc, n * T * H * W, Cc,
                     mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), coef.data_ptr(), 1, dz.data_ptr() + o * esz, None, lib.stream())
        if with_wgrad:
            lib.call("dpc_conv_wgrad", C.byref(dw), x_in.data_ptr() + o * esz, dz.data_ptr() + o * esz, Cc, part.data_ptr(), C.byref(ns), lib.stream())
        if with_dgrad:
            lib.call("dpc_conv_igemm", C.byref(dd), dz.data_ptr() + o * esz, wd.data_ptr(), dx.data_ptr() + o * esz, None, None, lib.stream())


def run(tag, k, reps=10, **kw):
    for _ in range(2):
        seq(k, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        seq(k, **kw)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:46s} chunks {k}: {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us per unit ({N * per * esz / k / 1e6:.0f} MB per tensor chunk)", flush=True)


print(f"tensor {N * per * esz / 1e6:.0f} MB, wgrad slabs {ns_full}")
for k in (1, 2, 4, 8, 16):
    run("apply alone", k, with_wgrad=False, with_dgrad=False)
for k in (1, 2, 4, 8, 16):
    run("dgrad alone", k, with_wgrad=False, with_apply=False)
for k in (1, 2, 4, 8):
    run("wgrad alone", k, with_wgrad=True, with_apply=False, with_dgrad=False)
for k in (1, 2, 4, 8, 16):
    run("apply -> dgrad", k, with_wgrad=False)
for k in (1, 2, 4, 8):
    run("apply -> wgrad -> dgrad", k, with_wgrad=True)
