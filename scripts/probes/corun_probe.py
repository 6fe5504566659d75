"""Does an input-gradient kernel compute the same bits when a weight-gradient kernel runs beside it on another stream?
The two-stream schedule diverged (rarely) when side work ran beside igemm_ws / igemm_wsp; this probe isolates
the pair: the input-gradient is launched ITERS times on the main stream while a weight gradient loops on a side stream, every
result is compared bit for bit with the result of a run without company, and the shape of a difference is printed (which rows /
columns, how large).  usage: python scripts/probes/corun_probe.py [l3|l2|l1] [iters]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "l3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N, T, H, W, Cc, ks, pd = {"l3": (1024, 3, 8, 8, 256, (3, 3, 3), (1, 1, 1)), "l2": (1024, 5, 16, 16, 128, (1, 3, 3), (0, 1, 1)),
                          "l1": (1024, 5, 32, 32, 64, (1, 3, 3), (0, 1, 1))}[which]
lib = L.Lib(os.environ["DPC_PROBE_LIB"], "hip") if os.environ.get("DPC_PROBE_LIB") else L.load_hip()   # e.g. scripts/probes/libdpc_nofix.so
dev = torch.device("cuda:0")
BF = torch.bfloat16
taps = ks[0] * ks[1] * ks[2]
g = torch.Generator(device="cuda").manual_seed(5)
dy = torch.randn(N, T, H, W, Cc, device=dev, generator=g).to(BF)
x = torch.randn(N, T, H, W, Cc, device=dev, generator=g).to(BF)
wd = (torch.randn(Cc, taps * Cc, device=dev, generator=g) * 0.05).to(BF)
out = torch.empty(N, T, H, W, Cc, device=dev, dtype=BF)
dd = kc.conv_desc(BF, BF, 1, N, (T, H, W), (T, H, W), Cc, Cc, Cc, taps * Cc, Cc, ks, (1, 1, 1), pd)
dw = kc.conv_desc(BF, torch.float32, 0, N, (T, H, W), (T, H, W), Cc, Cc, Cc, taps * Cc, Cc, ks, (1, 1, 1), pd)
ns = C.c_int32(0)
lib.call("dpc_conv_wgrad", C.byref(dw), None, None, Cc, None, C.byref(ns), lib.stream())
part = torch.zeros(ns.value, Cc, taps * Cc, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def dgrad():
    rc = lib.call("dpc_conv_igemm", C.byref(dd), dy, wd, out, None, None, C.c_void_p(main.cuda_stream))
    assert rc == 0


gw = torch.zeros(Cc, Cc, taps, device=dev)


def wgrad():   # weight gradient + the reduction of its split-K slabs, as the engine launches them (the reduction's small workgroups
    # are the only ones that fit on a CU beside a 144 KB input-gradient workgroup)
    rc = lib.call("dpc_conv_wgrad", C.byref(dw), x, dy, Cc, part, C.byref(ns), C.c_void_p(side.cuda_stream))
    assert rc == 0
    rc = lib.call("dpc_reduce_unpack", part, ns.value, gw, Cc, taps, Cc, Cc * taps, 1, taps, 0, C.c_void_p(side.cuda_stream))
    assert rc == 0


dgrad()
torch.cuda.synchronize()
print(which, "input-gradient:", L.last_kernel(lib), flush=True)
ref = out.clone()
for _ in range(20):   # alone: is it reproducible at all?
    out.zero_()
    dgrad()
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "not reproducible even without company"
wgrad()
torch.cuda.synchronize()
print(which, "weight gradient:", L.last_kernel(lib), flush=True)
bad = 0
for it in range(iters):
    out.zero_()
    torch.cuda.synchronize()
    wgrad()
    wgrad()
    dgrad()
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        d = (out.float() - ref.float()).view(-1, Cc)
        rows = (d.abs().amax(1) > 0).nonzero().flatten()
        cols = (d.abs().amax(0) > 0).nonzero().flatten()
        print(f"iteration {it}: {rows.numel()} rows differ (first {rows[:6].tolist()}, last {rows[-3:].tolist()}; span {int(rows[-1] - rows[0]) + 1}), "
              f"{cols.numel()} columns (first {cols[:4].tolist()} last {cols[-2:].tolist()}), max |diff| {d.abs().max().item():.4f} of max |ref| {ref.float().abs().max().item():.2f}",
              flush=True)
print(f"{which}: {bad} of {iters} launches differ from the run without company")
