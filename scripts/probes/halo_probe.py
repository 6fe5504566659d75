"""Timing probe of conv_halo_ws_kernel (layer1 conv of cfg2, forward and input-gradient with residual).  DPC_WS_PROBE build;
DPC_WS_DBG bits: 1 no patch DMA, 2 no MFMA loop, 4 no staging writes, 8 no epilogue (stores / BN sums)."""
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
N, T, H, W, Ci, Co = 1024, 5, 32, 32, 64, 64
d = kc.conv_desc(BF, BF, 0, N, (T, H, W), (T, H, W), Ci, Ci, Co, 9 * Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
src = torch.randn(N, T, H, W, Ci, device=dev).to(BF)
wgt = (torch.randn(Co, 9 * Ci, device=dev) * 0.05).to(BF)
out = torch.empty(N, T, H, W, Co, device=dev, dtype=BF)
rows = lib.call("dpc_conv_stats_rows", C.byref(d))
stats = torch.zeros(rows, 2, Co, device=dev)
flops = 2.0 * N * T * H * W * Co * 9 * Ci


def run(tag, dbg, reps=20):
    os.environ["DPC_WS_DBG"] = str(dbg)
    for _ in range(3):
        lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), wgt.data_ptr(), out.data_ptr(), None, stats.data_ptr(), lib.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), wgt.data_ptr(), out.data_ptr(), None, stats.data_ptr(), lib.stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{tag:50s} dbg={dbg:3d}  {us:8.1f} us  {flops / us * 1e-6:7.1f} TFLOP/s-equivalent", flush=True)


# round 5: what a BatchNorm-apply + ReLU of the source would cost inside this kernel (DPC_WS_DBG bit 256, see conv_halo.hip)
scratch = torch.empty_like(src)
_set = lib.c.dpc_probe_set_halo_buf
_set.argtypes, _set.restype = [C.c_void_p], C.c_int
_set(scratch.data_ptr())
run("full kernel", 0)
run("full kernel", 0)
run("BNIN load probe: patch rewritten in LDS + activation stored", 256)
run("BNIN load probe: patch rewritten in LDS, no activation store", 256 + 512)
run("full kernel", 0)
run("BNIN load probe: patch rewritten in LDS + activation stored", 256)
run("round-robin tile slots (no XCD grouping)", 16)
run("compute waves at default priority", 32)
run("full kernel", 0)
run("no patch DMA", 1)
run("no MFMA loop", 2)
run("no staging writes", 4)
run("no epilogue (stores, BN sums)", 8)
run("no epilogue, no staging", 12)
run("no DMA, no epilogue, no staging (MFMA + reads + barriers)", 13)
run("no MFMA, no staging (DMA + epilogue + barriers)", 6)
run("same, round-robin tile slots", 22)
run("half of the fragment reads left out (wrong results: what the loop costs with half its LDS reads)", 64)
run("  + no DMA, no epilogue, no staging", 64 + 13)
run("full kernel", 0)
