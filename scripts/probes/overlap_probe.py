"""Can a memory-bound BatchNorm-backward pass and a matrix-bound weight-gradient kernel share the GPU?  Launches
bn_bwd_reduce + bn_bwd_apply (layer1 tensor, 671 MB) on one stream and wgrad_patch (layer1 / layer2 shapes) on another,
and compares back-to-back with concurrent execution.  (Production library.)"""
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
dc = L.dtype_code(BF)


def make_wgrad(N, T, H, W, Ci, Co, ks, pd):
    taps = ks[0] * ks[1] * ks[2]
    d = kc.conv_desc(BF, torch.float32, 0, N, (T, H, W), (T, H, W), Ci, Ci, Co, taps * Ci, Co, ks, (1, 1, 1), pd)
    x = torch.randn(N, T, H, W, Ci, device=dev).to(BF)
    gy = torch.randn(N, T, H, W, Co, device=dev).to(BF)
    ns = C.c_int32(0)
    lib.call("dpc_conv_wgrad", C.byref(d), None, None, Co, None, C.byref(ns), lib.stream())
    part = torch.zeros(ns.value, Co, taps * Ci, device=dev)

    def run(stream):
        lib.call("dpc_conv_wgrad", C.byref(d), x.data_ptr(), gy.data_ptr(), Co, part.data_ptr(), C.byref(ns), stream)
    return run, (d, x, gy, part)


def make_bn(rows, Cc):
    dy = torch.randn(rows, Cc, device=dev).to(BF)
    raw = torch.randn(rows, Cc, device=dev).to(BF)
    mask = torch.randint(0, 256, (rows * Cc // 8,), device=dev, dtype=torch.uint8)
    mean = torch.zeros(Cc, device=dev); invstd = torch.ones(Cc, device=dev); gamma = torch.ones(Cc, device=dev)
    gw = torch.zeros(Cc, device=dev); gb = torch.zeros(Cc, device=dev)
    dx = torch.empty_like(dy)
    pr = C.c_int32(0)
    lib.call("dpc_bn_bwd_reduce", None, None, None, None, dc, rows, Cc, None, None, 1, None, C.byref(pr), lib.stream())
    stats = torch.zeros(max(pr.value, 1) * 2 * Cc, device=dev)
    coef = torch.zeros(2, Cc, device=dev)

    def run(stream):
        lib.call("dpc_bn_bwd_reduce", dy.data_ptr(), None, mask.data_ptr(), raw.data_ptr(), dc, rows, Cc, mean.data_ptr(), invstd.data_ptr(), 1,
                 stats.data_ptr(), C.byref(pr), stream)
        lib.call("dpc_bn_bwd_finalize", stats.data_ptr(), pr.value, Cc, float(rows), gw.data_ptr(), gb.data_ptr(), coef.data_ptr(), stream)
        lib.call("dpc_bn_bwd_apply", dy.data_ptr(), None, mask.data_ptr(), raw.data_ptr(), dc, rows, Cc, mean.data_ptr(), invstd.data_ptr(),
                 gamma.data_ptr(), coef.data_ptr(), 1, dx.data_ptr(), None, stream)
    def apply_only(stream):   # the reduction was taken in the producing input-gradient's epilogue (round 3): finalize + apply
        lib.call("dpc_bn_bwd_finalize", stats.data_ptr(), pr.value, Cc, float(rows), gw.data_ptr(), gb.data_ptr(), coef.data_ptr(), stream)
        lib.call("dpc_bn_bwd_apply", dy.data_ptr(), None, mask.data_ptr(), raw.data_ptr(), dc, rows, Cc, mean.data_ptr(), invstd.data_ptr(),
                 gamma.data_ptr(), coef.data_ptr(), 1, dx.data_ptr(), None, stream)
    run.apply_only = apply_only
    return run, (dy, raw, mask, dx, stats, coef)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


main = torch.cuda.current_stream()
side = torch.cuda.Stream()
for name, wg_args, rows, Cc in (("layer1", (1024, 5, 32, 32, 64, 64, (1, 3, 3), (0, 1, 1)), 1024 * 5 * 32 * 32, 64),
                                 ("layer2", (1024, 5, 16, 16, 128, 128, (1, 3, 3), (0, 1, 1)), 1024 * 5 * 16 * 16, 128),
                                 ("layer3", (1024, 3, 8, 8, 256, 256, (3, 3, 3), (1, 1, 1)), 1024 * 3 * 8 * 8, 256)):
    wg, keep1 = make_wgrad(*wg_args)
    bn, keep2 = make_bn(rows, Cc)
    ms = main.cuda_stream

    def both_seq():
        wg(ms); bn(ms)

    def both_par():
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        wg(side.cuda_stream)
        bn(ms)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)

    t_w = timeit(lambda: wg(ms)); t_b = timeit(lambda: bn(ms)); t_s = timeit(both_seq); t_p = timeit(both_par)
    print(f"{name}: wgrad {t_w:7.1f} us, bn backward (reduce+apply) {t_b:7.1f} us, back to back {t_s:7.1f} us, two streams {t_p:7.1f} us", flush=True)

    def a_seq():
        wg(ms); bn.apply_only(ms)

    def a_par():
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        wg(side.cuda_stream)
        bn.apply_only(ms)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)

    t_a = timeit(lambda: bn.apply_only(ms)); t_s = timeit(a_seq); t_p = timeit(a_par)
    print(f"{name}: wgrad {t_w:7.1f} us, finalize + apply {t_a:7.1f} us, back to back {t_s:7.1f} us, two streams {t_p:7.1f} us", flush=True)
