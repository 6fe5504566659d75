"""GPU tier: hipGraph replay of the whole train step == the same steps launched kernel by kernel; the gradient
exchange over RCCL with one rank (the driver's 1-GPU box) and -- when the node has >= 2 devices -- two ranks."""
import os
import subprocess
import sys

import pytest
import torch

from dpc_amd.engine import DPCEngine
from oracle import dpc_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eng(dtype, B=4, seed=233):
    eng = DPCEngine("resnet18", 64, 8, 5, 3, B, DEV, dtype, seed=seed)
    eng.load_params(O.init_params_reference_style("resnet18", seed=0))
    return eng


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_replay_equals_eager(dtype):
    x = torch.randn(4, 8, 3, 5, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    a, b = _eng(dtype), _eng(dtype)
    for _ in range(5):
        ra = a.train_step(x).clone()
    replay = b.capture_train_step(x, warmup=2)  # 2 real warm-up steps, then every replay is one more step
    assert len(replay.graphs) == 1
    from dpc_amd import engine as E_
    # one capture per (engine, input buffer, exchange): asking again returns the replay that exists, the graph list does not grow
    live = len(E_._LIVE_GRAPHS)
    assert b.capture_train_step(x, warmup=2) is replay and len(b._capture_graphs) == 1 and len(E_._LIVE_GRAPHS) == live
    for _ in range(3):
        rb = replay().clone()
    torch.cuda.synchronize()
    assert a.step_count == b.step_count == 5 and int(b.dev_step.item()) == 5
    assert torch.equal(a.flat_p, b.flat_p)  # deterministic kernels, same dropout stream (seed, step)
    assert torch.equal(ra, rb)
    # dropout masks change from step to step inside the graph (keyed on the device-side counter)
    m5 = b.dropout_masks_of_step()
    replay()
    m6 = b.dropout_masks_of_step()
    assert not torch.equal(m5, m6) and abs(m6.ne(0).float().mean().item() - 0.9) < 0.01
    # a refilled input buffer is picked up by the replay
    x.copy_(torch.randn(x.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(2)))
    r7 = replay().clone()
    assert not torch.equal(r7, rb)
    # lr / wd are kernel arguments of the captured Adam launch: a new learning rate is a new capture, the old replay refuses
    p7 = b.flat_p.clone()
    b.lr = 0.5e-3
    with pytest.raises(RuntimeError, match="capture_train_step"):
        replay()
    replay2 = b.capture_train_step(x, warmup=0)
    assert replay2 is not replay and len(b._capture_graphs) == 2
    replay2()
    torch.cuda.synchronize()
    assert not torch.equal(b.flat_p, p7)
    b.lr = 1e-3
    assert b.capture_train_step(x, warmup=0) is replay   # the first capture is still there for its own lr
    # dropped captures are parked until the process exits, never destroyed while it runs (engine._LIVE_GRAPHS says why)
    g0 = replay.graphs[0]
    b.release_captures()
    if os.environ.get("DPC_KEEP_GRAPHS", "1") != "0":
        assert not b._captures and any(g is g0 for g in E_._LIVE_GRAPHS) and len(E_._LIVE_GRAPHS) == live + 4   # two graphs + their events
    else:   # the A/B of scripts/gpu_r5_graph_destroy.sh: destroyed here, graphs first
        assert not b._captures and len(E_._LIVE_GRAPHS) == live and not replay.graphs


_RANK_SCRIPT = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from dpc_amd.engine import DPCEngine
from dpc_amd.parallel import make_allreduce, shard_of
from oracle import dpc_oracle as O
B = 2
xg = O.make_input_pcg(B * world, 8, 5, 64)
x = xg[shard_of(B * world, world, rank)].contiguous().to(dev)

def engine():
    e = DPCEngine("resnet18", 64, 8, 5, 3, B, dev, torch.float32, seed=233 + rank)
    e.load_params(O.make_params_pcg("resnet18"))
    return e

# (1) one step by hand, exactly as train_step does the two-bucket exchange; the local gradient is kept for the checks
eng = engine()
ones = torch.ones(eng.n_steps, eng.M, eng.D, device=dev)
ar = make_allreduce(dist, world, force=True)
eng.forward(x, train=True, dropout_masks=ones)
res = eng.loss_topk(True).clone()
local_tail = []
def on_tail(tail):
    local_tail.append(tail.clone())
    ar.start(tail)
eng.backward(on_tail_ready=on_tail)
local = torch.cat([eng.flat_g[:eng.grad_split].clone(), local_tail[0]])
ar.finish(eng.flat_g[:eng.grad_split])
avg = eng.flat_g.clone()
eng.adam_step()
# (2) the same 4 train steps (Philox dropout on) launched three ways must leave bit-identical parameters:
#     A eager + exchange | C three hipGraphs cut at the exchange points | (world 1 only) D one hipGraph, no exchange
outs = {{}}
for tag in ("A", "C") + (("D",) if world == 1 else ()):
    e = engine()
    a = make_allreduce(dist, world, force=True) if tag != "D" else None
    if tag == "A":
        for _ in range(4):
            r_ = e.train_step(x, allreduce=a).clone()
        ng = 0
    else:
        rp = e.capture_train_step(x, allreduce=a, warmup=2)
        for _ in range(2):
            r_ = rp().clone()
        ng = len(rp.graphs)
    torch.cuda.synchronize()
    outs[tag] = dict(params=e.flat_p.cpu(), res=r_.cpu(), ngraphs=ng, steps=e.step_count)
torch.save({{"local": local.cpu(), "avg": avg.cpu(), "params": eng.flat_p.cpu(), "res": res.cpu(), "modes": outs}},
           os.path.join({out!r}, f"rank{{rank}}.pt"))
dist.barrier()
dist.destroy_process_group()
"""


def _run_ranks(world, tmp_path, launcher):
    """the ranks as child processes -- forked by the pre-GPU launcher of conftest.py, not by this (GPU-initialised) process"""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
            "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), str(script)]
    if launcher is not None:
        rc, _, err = launcher.run(argv, env, 900)
    else:   # a single test run by hand without "-m gpu"
        r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=900)
        rc, err = r.returncode, r.stderr
    assert rc == 0, err[-4000:]
    return [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]


def _check_modes(r):
    m = r["modes"]
    assert m["C"]["ngraphs"] == 3 and m["A"]["steps"] == m["C"]["steps"] == 4
    assert torch.equal(m["A"]["params"], m["C"]["params"]) and torch.equal(m["A"]["res"], m["C"]["res"])
    assert torch.isfinite(m["C"]["res"]).all()


def test_rccl_single_rank_exchange_and_graph_cut(tmp_path, clean_launcher):
    """the RCCL path with world_size 1 (all a 1-GPU box can run): eager two-bucket exchange == three-graph replay cut at
    the exchange points == one graph without exchange, bit for bit"""
    (r,) = _run_ranks(1, tmp_path, clean_launcher)
    _check_modes(r)
    m = r["modes"]
    assert m["D"]["ngraphs"] == 1 and torch.equal(m["A"]["params"], m["D"]["params"])
    assert torch.equal(r["avg"], r["local"])  # averaging over one rank is the identity


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X on the node")
def test_rccl_two_rank_data_parallel(tmp_path, clean_launcher):
    """as tests/test_parallel_gloo.py, over RCCL/xGMI: averaged gradient == mean of the local ones, identical parameters
    on both ranks after the step (eager and graph-replayed), local gradients differ (per-rank shards, negatives and BN
    statistics, dpc/main.py:180,211-213)"""
    r = _run_ranks(2, tmp_path, clean_launcher)
    assert torch.equal(r[0]["params"], r[1]["params"]) and torch.equal(r[0]["avg"], r[1]["avg"])
    assert not torch.equal(r[0]["local"], r[1]["local"])
    assert torch.allclose(r[0]["avg"], 0.5 * (r[0]["local"] + r[1]["local"]), rtol=0, atol=1e-6)
    for i in range(2):
        _check_modes(r[i])
    assert torch.equal(r[0]["modes"]["C"]["params"], r[1]["modes"]["C"]["params"])
    p = O.make_params_pcg("resnet18")
    xg = O.make_input_pcg(4, 8, 5, 64)
    for i in range(2):
        loss, accs, grads, _ = O.train_step_reference(p, xg[2 * i:2 * i + 2], "resnet18", 3, None)
        assert abs(r[i]["res"][0].item() - loss.item()) < 1e-3
