"""How long does a fresh process take to reach its steady step time?  Replays the captured cfg2 train step N times and prints the
per-step time (HIP events around every replay) in blocks of 10, for the two-stream and the one-stream schedule.  Why: BENCH_r04's
`also.one_stream` leg (measured minutes into the process) was 0.35 % slower than the headline (measured 0.4 s into it), while an
alternating fresh-process A/B of the same two schedules says 3.3 % (profiles/r05_two_stream_ab.txt).
usage: python scripts/probes/clock_ramp_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import dpc_amd  # noqa: F401  (runtime defaults before HIP initialises)
from dpc_amd.engine import DPCEngine
from dpc_amd.model import DPC_RNN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
t_start = time.perf_counter()


def build(streams):
    os.environ["DPC_WGRAD_STREAM"] = str(streams)
    eng = DPCEngine("resnet18", 128, 8, 5, 3, 128, dev, torch.bfloat16, seed=233)
    init = DPC_RNN(128, network="resnet18", pred_step=3, seed=0)
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    block = torch.randn(128, 8, 3, 5, 128, 128, device=dev, generator=torch.Generator(dev).manual_seed(1234))
    return eng, eng.capture_train_step(block)


def trace(tag, step, n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter() - t_start
    ev[0].record()
    for i in range(n):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    blocks = [sum(ms[i:i + 10]) / len(ms[i:i + 10]) for i in range(0, n, 10)]
    print(f"{tag}: started {t0:.1f} s into the process; ms/step per block of 10: " + " ".join(f"{b:.2f}" for b in blocks), flush=True)
    return blocks


e2, s2 = build(1)
trace("two streams, cold", s2, n)
e1, s1 = build(0)
trace("one stream, after the above", s1, n)
trace("two streams, again", s2, 100)
trace("one stream, again", s1, 100)
time.sleep(5.0)
trace("two streams, after 5 s idle", s2, 100)
