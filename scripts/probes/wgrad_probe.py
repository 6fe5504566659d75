"""Timing probe of wgrad_patch_kernel (layer1 / layer2 / layer3 shapes of cfg2): which phase costs what.  Uses the
-DDPC_WS_PROBE build (scripts/probes/libdpc_probe.so); DPC_WS_DBG bits: 1 no DMA, 2 no compute, 4 no chunk barrier."""
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


def shape(name, N, T, H, W, Ci, Co, ks, pd):
    taps = ks[0] * ks[1] * ks[2]
    d = kc.conv_desc(BF, torch.float32, 0, N, (T, H, W), (T, H, W), Ci, Ci, Co, taps * Ci, Co, ks, (1, 1, 1), pd)
    x = torch.randn(N, T, H, W, Ci, device=dev).to(BF)
    gy = torch.randn(N, T, H, W, Co, device=dev).to(BF)
    ns = C.c_int32(0)
    os.environ["DPC_WS_DBG"] = "0"
    lib.call("dpc_conv_wgrad", C.byref(d), None, None, Co, None, C.byref(ns), lib.stream())
    part = torch.zeros(ns.value, Co, taps * Ci, device=dev)
    flops = 2.0 * N * T * H * W * Co * taps * Ci

    def run(tag, dbg, reps=20):
        os.environ["DPC_WS_DBG"] = str(dbg)
        for _ in range(3):
            lib.call("dpc_conv_wgrad", C.byref(d), x.data_ptr(), gy.data_ptr(), Co, part.data_ptr(), C.byref(ns), lib.stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.call("dpc_conv_wgrad", C.byref(d), x.data_ptr(), gy.data_ptr(), Co, part.data_ptr(), C.byref(ns), lib.stream())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:8s} nsplit={ns.value:3d} {tag:44s} dbg={dbg:2d} {us:8.1f} us {flops / us * 1e-6:7.1f} TFLOP/s-eq", flush=True)

    run("full", 0)
    run("full", 0)
    run("no DMA", 1)
    run("no compute (DMA + barriers)", 2)
    run("no DMA, no barrier (MFMA + reads)", 5)
    run("full", 0)


shape("layer1", 1024, 5, 32, 32, 64, 64, (1, 3, 3), (0, 1, 1))
shape("layer2", 1024, 5, 16, 16, 128, 128, (1, 3, 3), (0, 1, 1))
shape("layer3", 1024, 3, 8, 8, 256, 256, (3, 3, 3), (1, 1, 1))
