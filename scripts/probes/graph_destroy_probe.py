"""Does destroying captured train-step graphs corrupt host memory?  Repeats what tests/test_graph_rccl_gpu.py::test_graph_replay_equals_eager
does (two engines, eager steps, capture, replays, everything released) N times in one process and counts the iterations that fail --
the flake of the GPU tier was a KeyError out of a freshly built dict in DPCEngine.load_params, or a segfault.
usage: DPC_KEEP_GRAPHS=0|1 python scripts/probes/graph_destroy_probe.py [iterations]"""
import faulthandler
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpc_amd.engine import DPCEngine  # noqa: E402
from oracle import dpc_oracle as O  # noqa: E402

faulthandler.enable()
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for it in range(n):
    for dtype in (torch.float32, torch.bfloat16):
        try:
            x = torch.randn(4, 8, 3, 5, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
            engs = []
            for _ in range(2):
                e = DPCEngine("resnet18", 64, 8, 5, 3, 4, DEV, dtype)
                e.load_params(O.init_params_reference_style("resnet18", seed=0))
                engs.append(e)
            a, b = engs
            for _ in range(3):
                ra = a.train_step(x).clone()
            replay = b.capture_train_step(x, warmup=2)
            rb = replay().clone()
            torch.cuda.synchronize()
            assert torch.equal(ra, rb) and torch.equal(a.flat_p, b.flat_p)
            del a, b, engs, replay, e
        except Exception as ex:   # noqa: BLE001
            bad += 1
            print(f"iteration {it} {dtype}: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
        gc.collect()
print(f"DPC_KEEP_GRAPHS={os.environ.get('DPC_KEEP_GRAPHS', '1')}: {bad} of {2 * n} engine pairs failed", flush=True)
