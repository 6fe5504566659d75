#!/usr/bin/env python
"""HBM-bound kernel timing at layer1 size of cfg2 (5.24 M positions x 64 channels bf16 = 671 MB per tensor):
batch-norm apply / backward reduce / backward apply through the C ABI, HIP events, GB/s of the tensors moved.
Usage: python scripts/elt_bench.py [--rows N] [--C 64] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpc_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5242880)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = L.load_hip()
    dev = torch.device("cuda", 0)
    st = lib.stream()
    rows, Cc = a.rows, a.C
    bf = torch.bfloat16
    x = torch.randn(rows, Cc, device=dev).to(bf)
    res = torch.randn(rows, Cc, device=dev).to(bf)
    dy = torch.randn(rows, Cc, device=dev).to(bf)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    dz = torch.empty_like(x)
    mask = torch.zeros(rows * Cc // 8, dtype=torch.uint8, device=dev)
    f = lambda: torch.rand(Cc, device=dev) + 0.5  # noqa: E731
    scale, shift, mean, invstd, gamma = f(), f(), f(), f(), f()
    coef = torch.rand(2, Cc, device=dev) * 0.01
    pr = C.c_int32(0)
    lib.call("dpc_bn_bwd_reduce", None, None, None, None, 1, rows, Cc, None, None, 0, None, C.byref(pr), st)
    part = torch.empty(pr.value, 2, Cc, device=dev)
    T = rows * Cc * 2 / 1e9  # GB per tensor
    cases = [
        ("bn_apply            (2 T)", 2 + 1 / 16, lambda: lib.call("dpc_bn_apply", x, y, 1, rows, Cc, scale, shift, None, None, None, 1, mask, st)),
        ("bn_apply + residual (3 T)", 3 + 1 / 16, lambda: lib.call("dpc_bn_apply", x, y, 1, rows, Cc, scale, shift, res, None, None, 1, mask, st)),
        ("bn_bwd_reduce       (2 T)", 2 + 1 / 16, lambda: lib.call("dpc_bn_bwd_reduce", dy, None, mask, x, 1, rows, Cc, mean, invstd, 1, part, C.byref(pr), st)),
        ("bn_bwd_apply        (3 T)", 3 + 1 / 16, lambda: lib.call("dpc_bn_bwd_apply", dy, None, mask, x, 1, rows, Cc, mean, invstd, gamma, coef, 1, dx, None, st)),
        ("bn_bwd_apply + dz   (4 T)", 4 + 1 / 16, lambda: lib.call("dpc_bn_bwd_apply", dy, None, mask, x, 1, rows, Cc, mean, invstd, gamma, coef, 1, dx, dz, st)),
    ]
    def pair():
        lib.call("dpc_bn_bwd_reduce", dy, None, mask, x, 1, rows, Cc, mean, invstd, 1, part, C.byref(pr), st)
        return lib.call("dpc_bn_bwd_apply", dy, None, mask, x, 1, rows, Cc, mean, invstd, gamma, coef, 1, dx, None, st)
    cases.append(("reduce + apply      (5 T)", 5 + 2 / 16, pair))
    for name, nt, fn in cases:
        assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.iters
        print(f"{name}  {us:8.1f} us  {nt * T / us * 1e3:7.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
