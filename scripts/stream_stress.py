"""Are the two-stream and the one-stream schedule of the train step bit-identical, step after step?  Two engines of one
configuration (same parameters, same clip), one with the weight gradients on the side stream (default), one with
DPC_WGRAD_STREAM=0, both replaying their captured step; parameters and gradients are compared bit for bit after every step.
usage: python scripts/stream_stress.py [cfg2|cfg4|cfg5] [steps] [repeats] [modes, default 1,0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpc_amd.engine import DPCEngine  # noqa: E402
from dpc_amd.plan import CONFIGS  # noqa: E402
from oracle import dpc_oracle as O  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
modes = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("1", "0")   # DPC_WGRAD_STREAM of the two engines ("0,0": a control)
net, img, B = CONFIGS[cfg]
P = 5 if cfg == "cfg5" else 3
dev = "cuda:0"
bad = 0
for rep in range(repeats):
    engs = []
    for flag in modes:
        os.environ["DPC_WGRAD_STREAM"] = flag
        e = DPCEngine(net, img, 8, 5, P, B, dev, torch.bfloat16)
        e.load_params(O.init_params_reference_style(net, seed=rep))
        if os.environ.get("STRESS_GUARD"):   # out-of-bounds writes behind the split-K slab / statistics buffers would land in the guard
            for name in ("part", "stats"):
                t = getattr(e, name)
                big = torch.full((t.numel() + (1 << 22),), 12345.0, device=dev, dtype=t.dtype)
                setattr(e, name, big)
                setattr(e, "_guard_" + name, t.numel())
        engs.append(e)
    x = torch.randn(B, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(100 + rep))
    if os.environ.get("STRESS_EAGER"):   # kernel-by-kernel launches (stream events) instead of replaying captured graphs
        fns = [(lambda e=e: e.train_step(x)) for e in engs]
    else:
        fns = [e.capture_train_step(x) for e in engs]
    for s in range(steps):
        r = [f().clone() for f in fns]
        torch.cuda.synchronize()
        same_g = torch.equal(engs[0].flat_g, engs[1].flat_g)
        same_p = torch.equal(engs[0].flat_p, engs[1].flat_p)
        if os.environ.get("STRESS_GUARD") and s % 50 == 0:
            for e in engs:
                for name in ("part", "stats"):
                    t, n = getattr(e, name), getattr(e, "_guard_" + name)
                    if not bool((t[n:] == 12345.0).all()):
                        print(f"GUARD of {name} overwritten: {int((t[n:] != 12345.0).sum())} elements", flush=True)
        if not (same_g and same_p):
            bad += 1
            names = [k for k in engs[0].G if not torch.equal(engs[0].G[k], engs[1].G[k])]
            print(f"repeat {rep} step {s}: MISMATCH loss {r[0][0].item():.6f} vs {r[1][0].item():.6f}; gradients that differ ({len(names)}): {names}", flush=True)
            break
    else:
        print(f"repeat {rep}: {steps} steps bit-identical (loss {r[0][0].item():.5f})", flush=True)
    for e in engs:   # engine <-> unit reference cycles: collect before the next pair is built (two cfg5 engines are 180 GB)
        e.__dict__.clear()
    del engs, fns, e
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    print(f"  memory after repeat {rep}: {torch.cuda.memory_allocated() / 2**30:.1f} GiB allocated", flush=True)
print("MISMATCHES" if bad else "all identical")
sys.exit(1 if bad else 0)
