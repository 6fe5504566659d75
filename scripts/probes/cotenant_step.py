"""What does a co-tenant that owns k CUs cost the train step -- and which multi-GPU schedule should be the default?
(VERDICT r3, next #1b.)  RCCL's channel kernels during the gradient all-reduce are emulated on ONE GPU by the squatter
(csrc/diag.hip): k workgroups x 256 threads x 64 KB of LDS doing vector-memory traffic for `usec` microseconds on their own stream,
launched exactly where the exchange object of dpc_amd/parallel.py launches the all-reduce.

Schedules (the three-graph replay of DPCEngine.capture_train_step, as under torch.distributed):
  solo        no exchange at all (the 1-GPU step)
  overlap/0   tail all-reduce overlapped with layer1 + stem backward, persistent grids = all CUs (round 3's default)
  overlap/k   the same with DPC_RESERVE_CUS = k: persistent grids leave k CUs free while the tail is in flight
  serial      one all-reduce of the whole arena after the backward pass, nothing overlapped
usage: python scripts/probes/cotenant_step.py [cfg2] [usec=1000] [steps=20] [k list, default 8,16,32]"""
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpc_amd import _lib as L  # noqa: E402
from dpc_amd.engine import DPCEngine  # noqa: E402
from dpc_amd.model import DPC_RNN  # noqa: E402
from dpc_amd.plan import CONFIGS  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
usec = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ks = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [8, 16, 32]
c = CONFIGS[cfg]
net, img, P, batch = (c["net"], c["img_dim"], c["pred_step"], c["batch"]) if isinstance(c, dict) else (c[0], c[1], 3, c[2])
dev = torch.device("cuda:0")
lib = L.load_hip()
scratch = torch.randint(0, 1 << 30, (64 << 20,), device=dev, dtype=torch.int32)
sink = torch.zeros(4, device=dev, dtype=torch.int32)
comm = torch.cuda.Stream()


class SquatExchange:
    """start()/finish() like parallel.GradAllReduce: the 'all-reduce' is k squatter workgroups on the comm stream, ordered after the
    work already queued on the current stream (what RCCL does), joined in finish()."""

    def __init__(self, k, usec, two_bucket=True):
        self.k, self.usec = k, usec
        if two_bucket:
            self.start = self._start

    def _launch(self):
        cur = torch.cuda.current_stream()
        comm.wait_stream(cur)
        lib.call("dpc_diag_squat", self.k, 4, 64 * 1024, 2, self.usec, scratch, scratch.numel() * 4, None, sink, C.c_void_p(comm.cuda_stream))

    def _start(self, tail):
        self._launch()

    def finish(self, head):
        if not hasattr(self, "start"):
            self._launch()
        torch.cuda.current_stream().wait_stream(comm)

    def __call__(self, flat):
        self.finish(flat)


def measure(eng, block, exch):
    fn = eng.capture_train_step(block, allreduce=exch)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


eng = DPCEngine(net, img, 8, 5, P, batch, dev, torch.bfloat16, seed=233)
init = DPC_RNN(img, network=net, pred_step=P, seed=0)
eng.load_params({k: v.detach() for k, v in init.named_parameters()})
block = torch.randn(batch, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(1234))
rows = []
eng.reserve_cus = 0
solo = measure(eng, block, None)
rows.append(("solo", 0, solo))
print(f"{cfg}: solo {solo:.3f} ms/step; co-tenant = k x 256 threads x 64 KB LDS, vector-memory traffic, {usec} us", flush=True)
for k in ks:
    eng.reserve_cus = 0
    a = measure(eng, block, SquatExchange(k, usec))
    eng.reserve_cus = k
    b = measure(eng, block, SquatExchange(k, usec))
    eng.reserve_cus = 0
    s = measure(eng, block, SquatExchange(k, usec, two_bucket=False))
    rows += [(f"overlap/0 k={k}", k, a), (f"overlap/{k} k={k}", k, b), (f"serial k={k}", k, s)]
    print(f"k={k:3d}: overlap, grids=all CUs {a:.3f} ms ({100 * (a / solo - 1):+.1f} %) | overlap, {k} CUs reserved {b:.3f} ms ({100 * (b / solo - 1):+.1f} %)"
          f" | serial {s:.3f} ms ({100 * (s / solo - 1):+.1f} %)", flush=True)
print(json.dumps({"cfg": cfg, "usec": usec, "rows": [{"schedule": n, "k": k, "ms_per_step": round(v, 3)} for n, k, v in rows]}))
