"""Would the FORWARD pass gain from two streams?  (VERDICT r3 next #6d.)  The forward has no overlap today: LDS / MFMA-bound
convolutions alternate with HBM-bound BatchNorm applies on one stream.  With the batch cut into two halves A | B (BatchNorm still
over the whole batch: the halves' partial-sum rows feed one finalize), bn_apply_k(B) could run beside conv_{k+1}(A):

    one stream, whole batch :  apply_k            conv_{k+1}
    two halves              :  apply_k(A) | conv_{k+1}(A) || apply_k(B) | conv_{k+1}(B)

This probe measures the pieces with the existing kernels on two B = 64 engines and one B = 128 engine (cfg2 shapes): whole-batch
times, half-batch times, and the pair conv(A) || apply(B) on two streams; the projected gain per unit is
(apply + conv)_whole - (apply_half + pair + conv_half).  usage: python scripts/probes/fwd_overlap_probe.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpc_amd.engine import DPCEngine  # noqa: E402
from dpc_amd.model import DPC_RNN  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
init = DPC_RNN(128, network="resnet18", pred_step=3, seed=0)
prm = {k: v.detach() for k, v in init.named_parameters()}
engs = {}
for name, B in (("whole", 128), ("A", 64), ("B", 64)):
    os.environ["DPC_WGRAD_STREAM"] = "0"
    e = DPCEngine("resnet18", 128, 8, 5, 3, B, dev, torch.bfloat16)
    e.load_params(prm)
    x = torch.randn(B, 8, 3, 5, 128, 128, device=dev, generator=torch.Generator(dev).manual_seed(7))
    e.forward(x, train=True, materialise=False)   # every buffer (raw, act, masks) exists and holds plausible data
    engs[name] = e
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def pair(f1, f2):
    """f1 on stream 1 beside f2 on stream 2, reps times back to back; wall time per pair"""
    def go():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            f1()
        with torch.cuda.stream(s2):
            f2()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return timed(go)


def unit_of(e, bi, which):
    blk = e.blocks[bi]
    return blk, getattr(blk, which)


total_gain = 0.0
print(f"{'unit':22s} {'conv':>7s} {'apply':>7s} | {'conv/2':>7s} {'apply/2':>7s} {'pair':>7s} | {'one stream':>10s} {'two halves':>10s} {'gain':>6s}   (us)")
# (block index, unit, what the apply writes): the apply of unit k and the conv of the NEXT unit are what would overlap
plan = [(0, "c1"), (0, "c2"), (1, "c1"), (1, "c2"), (2, "c1"), (2, "c2"), (3, "c1"), (3, "c2"), (4, "c1"), (4, "c2"), (5, "c1"), (5, "c2")]
for bi, which in plan:
    t = {}
    for name in ("whole", "A", "B"):
        e = engs[name]
        blk, u = unit_of(e, bi, which)
        xin = blk.x_in if which == "c1" else blk.act1
        res = None if which == "c1" else (blk.ds.raw if blk.ds is not None else blk.x_in)
        ru = blk.ds if (which == "c2" and blk.ds is not None) else None
        out = blk.act1 if which == "c1" else blk.out
        relu = True if which == "c1" else blk.final_relu
        conv = (lambda e=e, u=u, xin=xin: e.call("dpc_conv_igemm", __import__("ctypes").byref(u.desc_f), xin, u.wp, u.raw, None, e.stats))
        app = (lambda u=u, out=out, relu=relu, res=res, ru=ru: u.apply(out, relu=relu, res=res, res_unit=ru))
        t[name] = (conv, app)
    cw, aw = timed(t["whole"][0]), timed(t["whole"][1])
    ch, ah = timed(t["A"][0]), timed(t["A"][1])
    pr = pair(t["A"][0], t["B"][1])
    one, two = cw + aw, ah + pr + ch
    total_gain += one - two
    print(f"blocks[{bi}].{which:14s} {cw:7.1f} {aw:7.1f} | {ch:7.1f} {ah:7.1f} {pr:7.1f} | {one:10.1f} {two:10.1f} {one - two:6.1f}")
print(f"projected forward gain over these {len(plan)} units: {total_gain / 1e3:.3f} ms per step (threshold for building it: 0.6 ms)")
