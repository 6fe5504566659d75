"""GPU tier: the two-stream schedule of the train step (weight gradients on a side stream beside the next unit's BatchNorm
backward, dpc_amd/engine.py: side()) is the SAME computation as the one-stream schedule: parameters and gradients bit for bit,
step after step, launched kernel by kernel and as a replayed hipGraph.  (scripts/stream_stress.py is the long-running form of this
check at the BASELINE batch sizes; profiles/r03_two_stream.txt holds its runs.)"""
import pytest
import torch

from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(monkeypatch, two_streams, net, size, B, P):
    monkeypatch.setenv("DPC_WGRAD_STREAM", "1" if two_streams else "0")
    e = DPCEngine(net, size, 8, 5, P, B, DEV, torch.bfloat16)
    e.load_params(O.init_params_reference_style(net, seed=3))
    assert (e._side is not None) == two_streams
    return e


@pytest.mark.parametrize("net,size,B,P,graph", [("resnet18", 128, 16, 3, False), ("resnet18", 128, 16, 3, True), ("resnet34", 224, 4, 3, True)])
def test_two_streams_bit_identical_to_one(monkeypatch, net, size, B, P, graph):
    a, b = build(monkeypatch, True, net, size, B, P), build(monkeypatch, False, net, size, B, P)
    x = torch.randn(B, 8, 3, 5, size, size, device=DEV, generator=torch.Generator(DEV).manual_seed(9))
    fa, fb = (a.capture_train_step(x), b.capture_train_step(x)) if graph else ((lambda: a.train_step(x)), (lambda: b.train_step(x)))
    for step in range(10):
        ra, rb = fa().clone(), fb().clone()
        torch.cuda.synchronize()
        assert torch.equal(ra, rb), (step, ra, rb)
        assert torch.equal(a.flat_g, b.flat_g), f"gradients differ at step {step}"
        assert torch.equal(a.flat_p, b.flat_p), f"parameters differ at step {step}"
    assert not a._busy   # every fork was joined
