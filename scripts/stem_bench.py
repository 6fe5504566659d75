#!/usr/bin/env python
"""Stem backward at cfg2 size (1024 clips x 5 frames, 64 x 64 space-to-depth grid, 64 channels, bf16), HIP events:
two-kernel form (dpc_pool_bn_bwd_apply -> dz tensor -> dpc_conv_wgrad) against dpc_stem_wgrad_fused; forward tail
(dpc_bn_relu_maxpool_fwd).  Usage: python scripts/stem_bench.py [--bn 1024] [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpc_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bn", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--probe", action="store_true", help="scripts/probes/libdpc_probe.so ('make probe'): DPC_SF_DBG leaves phases out")
    a = ap.parse_args()
    lib = L.Lib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libdpc_probe.so"), "hip") if a.probe else L.load_hip()
    dev = torch.device("cuda", 0)
    st = lib.stream()
    BN, T, H, Co = a.bn, 5, a.hw, 64
    bf = torch.bfloat16
    Ho = (H - 1) // 2 + 1
    xs = torch.randn(BN, T, H, H, 16, device=dev).to(bf)
    raw = torch.randn(BN, T, H, H, Co, device=dev).to(bf)
    dz = torch.empty_like(raw)
    pooled = torch.empty(BN * T, Ho, Ho, Co, device=dev, dtype=bf)
    am = torch.empty(BN * T, Ho, Ho, Co, dtype=torch.uint8, device=dev)
    gy = torch.randn(BN * T, Ho, Ho, Co, device=dev).to(bf)
    f = lambda: torch.rand(Co, device=dev) + 0.5  # noqa: E731
    scale, shift, mean, invstd, gamma = f(), f() - 1.0, f(), f(), f()
    coef = torch.rand(2, Co, device=dev) * 0.01
    d = L.ConvDesc(1, 0, 0, BN, T, H, H, T, H, H, 16, 16, Co, 256, Co, 1, 4, 4, 1, 1, 1, 0, 2, 2)
    ns = C.c_int32(0)
    lib.call("dpc_conv_wgrad", C.byref(d), None, None, Co, None, C.byref(ns), st)
    part = torch.zeros(ns.value, Co, 256, device=dev)
    ns2 = C.c_int32(0)
    lib.call("dpc_stem_wgrad_fused", C.byref(d), None, None, None, None, None, None, None, None, None, C.byref(ns2), st)
    part2 = torch.zeros(ns2.value, Co, 256, device=dev)
    dc = 1
    cases = [
        ("bn_relu_maxpool_fwd", lambda: lib.call("dpc_bn_relu_maxpool_fwd", raw, dc, BN * T, H, H, Co, scale, shift, pooled, am, st)),
        ("pool_bn_bwd_apply (writes dz)", lambda: lib.call("dpc_pool_bn_bwd_apply", gy, am, raw, dc, BN * T, H, H, Co, mean, invstd, gamma, coef, dz, st)),
        ("conv_wgrad (reads dz)", lambda: lib.call("dpc_conv_wgrad", C.byref(d), xs, dz, Co, part, C.byref(ns), st)),
        ("stem_wgrad_fused", lambda: lib.call("dpc_stem_wgrad_fused", C.byref(d), xs, raw, gy, am, mean, invstd, gamma, coef, part2, C.byref(ns2), st)),
    ]
    if a.probe:
        cases = cases[-1:]
    for name, fn in cases:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:34s} {1e3 * e0.elapsed_time(e1) / a.iters:9.1f} us  [{L.last_kernel(lib)}]", flush=True)
    if not a.probe:
        print("fused == two-kernel:", torch.equal(part, part2))


if __name__ == "__main__":
    main()
