"""Timing probe of the layer1 fused input-gradients (conv_halo_ws_kernel<HAS_ADD, 8, 128, EPI = true>: gated residual addend and / or
the BatchNorm-backward partial sums of the unit upstream in the epilogue) -- which phase binds them?  DPC_WS_PROBE build;
DPC_WS_DBG bits: 1 no patch DMA, 2 no MFMA loop, 4 no staging writes, 8 no epilogue (stores / sums), 64 half the fragment reads."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

lib = L.Lib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdpc_probe.so"), "hip")
dev = torch.device("cuda:0")
BF = torch.bfloat16
N, T, H, W, Cc = 1024, 5, 32, 32, 64
d = kc.conv_desc(BF, BF, 1, N, (T, H, W), (T, H, W), Cc, Cc, Cc, 9 * Cc, Cc, (1, 3, 3), (1, 1, 1), (0, 1, 1))   # mode 1: input-gradient
draw = torch.randn(N, T, H, W, Cc, device=dev).to(BF)
wd = (torch.randn(Cc, 9 * Cc, device=dev) * 0.05).to(BF)
dx = torch.empty(N, T, H, W, Cc, device=dev, dtype=BF)
addend = torch.randn(N, T, H, W, Cc, device=dev).to(BF)
raw = torch.randn(N, T, H, W, Cc, device=dev).to(BF)
nu = N * T * H * W * Cc // 8
amask = torch.randint(0, 256, (nu,), device=dev, dtype=torch.uint8)
bmask = torch.randint(0, 256, (nu,), device=dev, dtype=torch.uint8)
mean, invstd = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
rows = lib.call("dpc_conv_stats_rows", C.byref(d))
stats = torch.zeros(max(rows, 1), 2, Cc, device=dev)
flops = 2.0 * N * T * H * W * Cc * 9 * Cc
T1 = N * T * H * W * Cc * 2 / 1e9


def variant(name):
    ep = L.ConvEpilogue()
    if name in ("add", "add+red"):
        ep.addend, ep.addend_mask = addend.data_ptr(), amask.data_ptr()
    if name in ("red", "add+red"):
        ep.bn_raw, ep.bn_mask, ep.bn_mean, ep.bn_invstd, ep.stats = raw.data_ptr(), bmask.data_ptr(), mean.data_ptr(), invstd.data_ptr(), stats.data_ptr()
    return ep


def run(tag, name, dbg, reps=20):
    os.environ["DPC_WS_DBG"] = str(dbg)
    ep = variant(name)

    def go():
        if name == "plain":
            lib.call("dpc_conv_igemm", C.byref(d), draw.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, None, lib.stream())
        else:
            lib.call("dpc_conv_igemm_ex", C.byref(d), draw.data_ptr(), wd.data_ptr(), dx.data_ptr(), C.byref(ep), lib.stream())
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    nt = {"plain": 2, "add": 3.06, "red": 3.06, "add+red": 4.12}[name]
    print(f"{name:8s} {tag:58s} dbg={dbg:3d} {us:8.1f} us  {flops / us * 1e-6:7.1f} TFLOP/s-eq  {nt * T1 / us * 1e3:6.2f} TB/s-eq ({L.last_kernel(lib)})", flush=True)


for name in ("plain", "red", "add+red"):
    run("full kernel", name, 0)
    run("full kernel", name, 0)
    run("no patch DMA", name, 1)
    run("no MFMA loop", name, 2)
    run("no epilogue (loads of the fused pieces stay)", name, 8)
    run("no MFMA, no staging (DMA + epilogue + barriers)", name, 6)
    run("no DMA, no epilogue, no staging (MFMA + reads + barriers)", name, 13)
    run("no DMA, no MFMA, no staging: the helper waves' epilogue alone", name, 7)
    run("half of the fragment reads", name, 64)
    run("full kernel", name, 0)
