"""Deterministic repro of round 3's "LDS race" and the proof of its fix (VERDICT r3, next #1a).

The layer3 input-gradient (igemm_ws_kernel<false,false>: 215 VGPRs, 144 KB of LDS -- a small workgroup fits beside it) runs ITERS
times on the main stream while a squatter (csrc/diag.hip) is resident on every CU, once per (library, squatter) pair:

  library   nofix = scripts/probes/libdpc_nofix.so  (make nofix: the end-of-tile fragment reads are NOT retired -- round 3's code
                    without the 160 KB LDS claim that hid it)
            fixed = dpc_amd/libdpc_hip.so            (WS_RETIRE_TAIL_READS, no LDS claim)
  squatter  none | 8 KB LDS idle | 8 KB LDS + LDS traffic | no LDS + vector-memory traffic | no LDS + VALU | no LDS idle

Every result is compared bit for bit with a solo run.  Expected if the mechanism in conv_igemm_ws.hip is right: nofix fails
under ANY co-tenant that stretches the LDS round trip of the compute waves -- LDS or not (the full-LDS claim was not a fix) --
and the shape of a failure is one wave's 64 x 128 block (+ stray rows); fixed never fails.
usage: python scripts/probes/squat_probe.py [iters] [l3|l2]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kcases as kc  # noqa: E402
from dpc_amd import _lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
which = sys.argv[2] if len(sys.argv) > 2 else "l3"
N, T, H, W, Cc, ks, pd = {"l3": (1024, 3, 8, 8, 256, (3, 3, 3), (1, 1, 1)), "l2": (1024, 5, 16, 16, 128, (1, 3, 3), (0, 1, 1))}[which]
dev = torch.device("cuda:0")
BF = torch.bfloat16
taps = ks[0] * ks[1] * ks[2]
g = torch.Generator(device="cuda").manual_seed(5)
dy = torch.randn(N, T, H, W, Cc, device=dev, generator=g).to(BF)
wd = (torch.randn(Cc, taps * Cc, device=dev, generator=g) * 0.05).to(BF)
out = torch.empty(N, T, H, W, Cc, device=dev, dtype=BF)
dd = kc.conv_desc(BF, BF, 1, N, (T, H, W), (T, H, W), Cc, Cc, Cc, taps * Cc, Cc, ks, (1, 1, 1), pd)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
scratch = torch.randint(0, 1 << 30, (64 << 20,), device=dev, dtype=torch.int32)   # 256 MB: misses the L2
where = torch.zeros(1024, device=dev, dtype=torch.int32)
sink = torch.zeros(4, device=dev, dtype=torch.int32)

libs = {"fixed": L.load_hip()}
nofix = os.path.join(ROOT, "scripts", "probes", "libdpc_nofix.so")
if os.path.exists(nofix):
    libs = {"nofix": L.Lib(nofix, "hip"), "fixed": libs["fixed"]}
sq = libs["fixed"]

# (name, workgroups, waves, LDS bytes, mode).  ONE workgroup of four waves per CU (a wave on every SIMD: 2 x 216 + 48 VGPRs fit; two
# 2-wave workgroups can land on the same two SIMDs, 96 VGPRs there, and igemm_ws<false> -- 432 per SIMD -- then waits for them to
# leave: session A of round 4 measured 1 530 us per launch = squatter life + solo time, i.e. no co-residency at all)
SQUATS = [("none", 0, 0, 0, 0), ("lds8k-idle", 256, 4, 8192, 0), ("lds8k-ldstraffic", 256, 4, 8192, 1), ("nolds-vmem", 256, 4, 0, 2),
          ("nolds-valu", 256, 4, 0, 3), ("nolds-idle", 256, 4, 0, 0)]


def dgrad(lib):
    rc = lib.call("dpc_conv_igemm", C.byref(dd), dy, wd, out, None, None, C.c_void_p(main.cuda_stream))
    assert rc == 0


def squat(n, waves, lds, mode, usec, stream):
    sq.call("dpc_diag_squat", n, waves, lds, mode, usec, scratch, scratch.numel() * 4, where, sink, C.c_void_p(stream.cuda_stream))


dgrad(libs["fixed"])
torch.cuda.synchronize()
print(which, "input-gradient:", L.last_kernel(libs["fixed"]), flush=True)
ref = out.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dgrad(libs["fixed"]); e1.record(); torch.cuda.synchronize()
t_solo = e0.elapsed_time(e1) * 1e3
usec = int(t_solo * 2.5)
print(f"solo launch {t_solo:.0f} us; squatters live {usec} us", flush=True)

# placement of the squatters on an idle chip
squat(256, 4, 8192, 0, 50, main)
torch.cuda.synchronize()
w = where[:256].cpu().numpy().astype("uint32")
cus = {(int(v >> 16) & 0xf, int(v >> 13) & 0x7, int(v >> 12) & 1, int(v >> 8) & 0xf) for v in w if v >> 31}
print(f"256 squatter workgroups reported {len(cus)} distinct (xcc, se, sh, cu) placements", flush=True)

for lname, lib in libs.items():
    dgrad(lib)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), f"{lname}: solo result differs from the product library's"
    for sname, n, waves, lds, mode in SQUATS:
        bad, shapes, ms = 0, [], 0.0
        for it in range(iters):
            out.zero_()
            torch.cuda.synchronize()
            if n:
                squat(n, waves, lds, mode, usec, side)
                squat(1, 1, 0, 0, 30, main)      # 30 us head start: the squatters are resident when the input-gradient arrives
            e0.record()
            dgrad(lib)
            e1.record()
            torch.cuda.synchronize()
            ms += e0.elapsed_time(e1)
            if not torch.equal(out, ref):
                bad += 1
                if len(shapes) < 4:
                    d = (out.float() - ref.float()).view(-1, Cc)
                    rows = (d.abs().amax(1) > 0).nonzero().flatten()
                    cols = (d.abs().amax(0) > 0).nonzero().flatten()
                    shapes.append(f"{rows.numel()} rows x {cols.numel()} cols (rows {rows[:3].tolist()}..{rows[-2:].tolist()})")
        co = "shared the CUs" if (not n or ms / iters * 1e3 < 0.6 * (t_solo + usec)) else "did NOT co-reside (launch waited for the squatters)"
        print(f"{lname:6s} beside {sname:18s}: {bad:4d} of {iters} launches differ from the solo result; {ms / iters * 1e3:6.0f} us per launch, {co}"
              + (("   e.g. " + "; ".join(shapes)) if shapes else ""), flush=True)
