"""Timing probe of igemm_wsp_kernel (layer2 shape): which phase of the loop costs what.  Uses a -DDPC_WS_PROBE build of
conv_igemm_ws.hip (scripts/probes/libdpc_probe.so); DPC_WS_DBG bits leave phases out (results are then wrong by design)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

SHAPE = os.environ.get("WS_PROBE_SHAPE", "layer2")
lib = L.Lib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdpc_probe.so"), "hip")
dev = torch.device("cuda:0")
BF = torch.bfloat16
if SHAPE == "layer3":   # igemm_ws_kernel: 3x3x3, 8x8 planes, 256 -> 256 (two column tiles, temporally grouped tiles)
    N, T, H, W, Ci, Co, KS, PD = 1024, 3, 8, 8, 256, 256, (3, 3, 3), (1, 1, 1)
else:                   # igemm_wsp_kernel: 1x3x3, 16x16 planes, 128 -> 128
    N, T, H, W, Ci, Co, KS, PD = 1024, 5, 16, 16, 128, 128, (1, 3, 3), (0, 1, 1)
TAPS = KS[0] * KS[1] * KS[2]
if SHAPE in ("l3s", "l4s"):   # igemm_ws_kernel<false,true>: strided input-gradient by parity classes (layer3.0 / layer4.0 conv1)
    N, T, H, W, Ci, Co = (1024, 5, 16, 16, 128, 256) if SHAPE == "l3s" else (1024, 3, 8, 8, 256, 256)
    R = tuple((i + 2 - 3) // 2 + 1 for i in (T, H, W))
    d = kc.conv_desc(BF, BF, 1, N, (T, H, W), R, Co, Co, Ci, TAPS * Co, Ci, KS, (2, 2, 2), PD)
    src = torch.randn(N, *R, Co, device=dev).to(BF)            # dy
    wgt = (torch.randn(Ci, TAPS * Co, device=dev) * 0.05).to(BF)
    out = torch.empty(N, T, H, W, Ci, device=dev, dtype=BF)
    stats = None
    flops = 2.0 * N * R[0] * R[1] * R[2] * Co * TAPS * Ci
else:
    d = kc.conv_desc(BF, BF, 0, N, (T, H, W), (T, H, W), Ci, Ci, Co, TAPS * Ci, Co, KS, (1, 1, 1), PD)
    src = torch.randn(N, T, H, W, Ci, device=dev).to(BF)
    wgt = (torch.randn(Co, TAPS * Ci, device=dev) * 0.05).to(BF)
    out = torch.empty(N, T, H, W, Co, device=dev, dtype=BF)
    rows = lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = torch.zeros(rows, 2, Co, device=dev)
    flops = 2.0 * N * T * H * W * Co * TAPS * Ci


def run(tag, dbg, plane=True, reps=20):
    os.environ["DPC_WS_DBG"] = str(dbg)
    for _ in range(3):
        lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), wgt.data_ptr(), out.data_ptr(), None, stats.data_ptr() if stats is not None else None, lib.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.call("dpc_conv_igemm", C.byref(d), src.data_ptr(), wgt.data_ptr(), out.data_ptr(), None, stats.data_ptr() if stats is not None else None, lib.stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{tag:58s} dbg={dbg:3d}  {us:8.1f} us  {flops / us * 1e-6:7.1f} TFLOP/s-equivalent", flush=True)


run("full kernel (first timing of the process: clocks still settling)", 0)
run("full kernel", 0)
run("no global stores / stats in the epilogue", 1)
run("no epilogue", 2)
run("no DMA (loaders only keep the barriers)", 4)
run("no DMA, no epilogue", 6)
run("no patch DMA, no epilogue", 130)
run("no weight DMA, no epilogue", 258)
run("patches of 8 planes only (cache-resident source), no epilogue", 514)
run("no DMA, no epilogue, no barriers (MFMA + fragment reads)", 38)
run("full kernel again", 0)
