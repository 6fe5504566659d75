"""would half-batch software pipelining of the FORWARD pay -- bn_apply of one half beside the next convolution of the
other half?  The pairs it would create, in isolation, at cfg5 / cfg2 sizes: the MFMA-bound loader / compute convolution (igemm_ws /
igemm_wsp / conv_halo_ws) on one stream, the HBM-bound bn_apply (same layer's tensor size, half batch each) on the other; back to
back against concurrent.  (Production library.)"""
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


def make_conv(N, T, H, W, Ci, Co, ks, pd):
    taps = ks[0] * ks[1] * ks[2]
    d = kc.conv_desc(BF, BF, 0, N, (T, H, W), (T, H, W), Ci, Ci, Co, taps * Ci, Co, ks, (1, 1, 1), pd)
    x = torch.randn(N, T, H, W, Ci, device=dev).to(BF)
    w = (torch.randn(Co, taps * Ci, device=dev) * 0.05).to(BF)
    y = torch.empty(N, T, H, W, Co, device=dev, dtype=BF)
    rows = lib.call("dpc_conv_stats_rows", C.byref(d))
    stats = torch.zeros(max(rows, 1), 2, Co, device=dev)

    def run(stream):
        lib.call("dpc_conv_igemm", C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, stats.data_ptr(), stream)
    return run, (d, x, w, y, stats)


def make_apply(rows, Cc):
    raw = torch.randn(rows, Cc, device=dev).to(BF)
    y = torch.empty_like(raw)
    mask = torch.empty(rows * Cc // 8, device=dev, dtype=torch.uint8)
    sc, sh = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)

    def run(stream):
        lib.call("dpc_bn_apply", raw.data_ptr(), y.data_ptr(), dc, rows, Cc, sc.data_ptr(), sh.data_ptr(), None, None, None, 1, mask.data_ptr(), stream)
    return run, (raw, y, mask, sc, sh)


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
# half batches: cfg5 B = 64 -> 256 clips x blocks per half; cfg2 B = 128 -> 512
for name, cv, rows, Cc in (("cfg5 layer3 3x3x3 256->256 @14x14x3 (igemm_ws)", (256, 3, 14, 14, 256, 256, (3, 3, 3), (1, 1, 1)), 256 * 3 * 14 * 14, 256),
                           ("cfg5 layer2 1x3x3 128->128 @28x28 (igemm_ws)", (256, 5, 28, 28, 128, 128, (1, 3, 3), (0, 1, 1)), 256 * 5 * 28 * 28, 128),
                           ("cfg5 layer1 1x3x3 64->64 @56x56 (conv_halo_ws)", (256, 5, 56, 56, 64, 64, (1, 3, 3), (0, 1, 1)), 256 * 5 * 56 * 56, 64),
                           ("cfg2 layer3 3x3x3 256->256 @8x8x3 (igemm_ws)", (512, 3, 8, 8, 256, 256, (3, 3, 3), (1, 1, 1)), 512 * 3 * 8 * 8, 256),
                           ("cfg2 layer2 1x3x3 128->128 @16x16 (igemm_wsp)", (512, 5, 16, 16, 128, 128, (1, 3, 3), (0, 1, 1)), 512 * 5 * 16 * 16, 128),
                           ("cfg2 layer1 1x3x3 64->64 @32x32 (conv_halo_ws)", (512, 5, 32, 32, 64, 64, (1, 3, 3), (0, 1, 1)), 512 * 5 * 32 * 32, 64)):
    conv, keep1 = make_conv(*cv)
    bn, keep2 = make_apply(rows, Cc)
    ms = main.cuda_stream

    def seq():
        conv(ms); bn(ms)

    def par():
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        bn(side.cuda_stream)
        conv(ms)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)

    t_c = timeit(lambda: conv(ms)); k = L.last_kernel(lib); t_b = timeit(lambda: bn(ms)); t_s = timeit(seq); t_p = timeit(par)
    print(f"{name:52s} conv {t_c:7.1f} us ({k[:34]}), bn_apply {t_b:6.1f} us, back to back {t_s:7.1f} us, two streams {t_p:7.1f} us  "
          f"=> saves {t_s - t_p:6.1f} us = {100 * (t_s - t_p) / t_s:4.1f} % of the pair", flush=True)
    del conv, bn, keep1, keep2
    torch.cuda.empty_cache()
