#!/usr/bin/env python
"""Kernel-level timing of the conv shapes of configs[1] (resnet18, 128 px, batch 128/GPU, bf16):
forward implicit GEMM (dpc_conv_igemm) and weight gradient (dpc_conv_wgrad) through the C ABI,
HIP events on the launch stream, random data.  Usage: python scripts/conv_bench.py [name ...] [--iters N]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpc_amd import _lib as L  # noqa: E402

BF = torch.bfloat16
# name: (frames N, Ci, Co, T, H, W (source), kernel, stride, pad)
SHAPES = {
    "l1": (5120, 64, 64, 1, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "l2": (5120, 128, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "l2s": (5120, 64, 128, 1, 32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    "l3": (1024, 256, 256, 3, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "l3s": (1024, 128, 256, 5, 16, 16, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    "l2k": (1707, 128, 128, 3, 16, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),   # l2 geometry with a long reduction
    "l3k": (1024, 256, 256, 3, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # l3 geometry with a short one
    "l2c": (5120, 256, 128, 1, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # l2 with Ci=256
    "gru": (896, 256, 768, 1, 4, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0)),         # ConvGRU gate GEMMs batched over 7 steps (weight gradient)
    "gru1": (128, 256, 768, 1, 4, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0)),        # one step (forward / input gradient)
    "l4": (1024, 256, 256, 2, 4, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "l4s": (1024, 256, 256, 3, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    "l2s224": (1760, 64, 128, 1, 56, 56, (1, 3, 3), (1, 2, 2), (0, 1, 1)),   # cfg4 shard (44 x 8 clips x 5 frames)
    "l3s224": (352, 128, 256, 5, 28, 28, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
}


def odim(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["l1", "l2", "l2s", "l3", "l3s", "l4"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    args = ap.parse_args()
    lib = L.Lib(os.environ["DPC_BENCH_LIB"], "hip") if os.environ.get("DPC_BENCH_LIB") else L.load_hip()   # A/B against another build
    dev = torch.device("cuda", 0)
    st = lib.stream()
    for name in args.names:
        N, Ci, Co, T, H, W, ks, sd, pd = SHAPES[name]
        R = tuple(odim(i, k, s, p) for i, k, s, p in zip((T, H, W), ks, sd, pd))
        taps = ks[0] * ks[1] * ks[2]
        M = N * R[0] * R[1] * R[2]
        flops = 2.0 * M * taps * Ci * Co
        src = torch.randn(N, T, H, W, Ci, device=dev).to(BF)
        dy = torch.randn(N, *R, Co, device=dev).to(BF)
        wgt = (torch.randn(Co, taps * Ci, device=dev) * 0.05).to(BF)
        wgt_d = (torch.randn(Ci, taps * Co, device=dev) * 0.05).to(BF)
        out = torch.empty(N, *R, Co, device=dev, dtype=BF)
        dx = torch.empty(N, T, H, W, Ci, device=dev, dtype=BF)
        d = L.ConvDesc(1, 1, 0, N, *R, T, H, W, Ci, Ci, Co, taps * Ci, Co, *ks, *sd, *pd)
        dd = L.ConvDesc(1, 1, 1, N, T, H, W, *R, Co, Co, Ci, taps * Co, Ci, *ks, *sd, *pd)
        dw = L.ConvDesc(1, 0, 0, N, *R, T, H, W, Ci, Ci, Co, taps * Ci, Co, *ks, *sd, *pd)
        rows = lib.call("dpc_conv_stats_rows", C.byref(d))
        stats = torch.zeros(rows, 2, Co, device=dev)
        ns = C.c_int32(0)
        lib.call("dpc_conv_wgrad", C.byref(dw), None, None, Co, None, C.byref(ns), st)
        part = torch.empty(ns.value, Co, taps * Ci, device=dev)

        def run(op):
            if op == "fwd":
                return lib.call("dpc_conv_igemm", C.byref(d), src, wgt, out, None, stats, st)
            if op == "dgrad":
                return lib.call("dpc_conv_igemm", C.byref(dd), dy, wgt_d, dx, None, None, st)
            if op == "dgrada":  # input-gradient with a residual addend (first conv of a BasicBlock)
                return lib.call("dpc_conv_igemm", C.byref(dd), dy, wgt_d, dx, src, None, st)
            return lib.call("dpc_conv_wgrad", C.byref(dw), src, dy, Co, part, C.byref(ns), st)

        for op in args.ops.split(","):
            rc = run(op)
            assert rc == 0, (name, op, rc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run(op)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print(f"{name:4s} {op:6s} M={M:8d} K={taps * Ci:5d} Co={Co:4d}  {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
