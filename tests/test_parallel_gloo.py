"""Multi-rank path on CPU: world_size 2 over gloo, one process per rank, each rank running the
engine (kernels on the host SIMT simulator) on ITS shard; the flat gradient arena is averaged with
one all-reduce (dpc_amd/parallel.py) and Adam steps.  Checks the data-parallel contract of
SURVEY.md §8e: per-rank negatives / BN statistics / loss (dpc/main.py:180,211-213), averaged
gradients == mean of the per-shard reference gradients, identical parameters on all ranks after
the step."""
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WIDTHS = (8, 16, 32, 32)
SIZE, BPER = 64, 1


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpc_amd import _lib as L
    from dpc_amd.engine import DPCEngine
    from dpc_amd.parallel import make_allreduce, shard_of
    from oracle import dpc_oracle as O
    eng = DPCEngine("resnet18", SIZE, 8, 5, 3, BPER, "cpu", torch.float32, WIDTHS, lib=L.load_emulator())
    p = O.make_params_pcg("resnet18", WIDTHS)
    eng.load_params(p)
    x_global = O.make_input_pcg(BPER * world, 8, 5, SIZE)
    x = x_global[shard_of(BPER * world, world, rank)].contiguous()
    ones = torch.ones(eng.n_steps, eng.M, eng.D)
    eng.forward(x, train=True, dropout_masks=ones)
    res = eng.loss_topk(True).clone()
    # two-bucket exchange exactly as train_step does it: the tail all-reduce starts (asynchronously) while the
    # backward of layer1 + stem is still to run; the copy of the local tail is taken inside the callback
    ar = make_allreduce(dist, world)
    local_tail = []

    def on_tail(tail):
        local_tail.append(tail.clone())
        ar.start(tail)

    eng.backward(on_tail_ready=on_tail)
    local_grad = torch.cat([eng.flat_g[:eng.grad_split].clone(), local_tail[0]])
    ar.finish(eng.flat_g[:eng.grad_split])
    eng.adam_step()
    torch.save({"local_grad": local_grad, "avg_grad": eng.flat_g.clone(), "params": eng.flat_p.clone(), "res": res,
                "offsets": eng.offsets}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel(tmp_path):
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    world, port = 2, 29600 + os.getpid() % 300
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    # all-reduce averaged the arenas; both ranks hold the same gradient and the same updated parameters
    assert torch.equal(r[0]["avg_grad"], r[1]["avg_grad"])
    assert torch.equal(r[0]["params"], r[1]["params"])
    assert torch.allclose(r[0]["avg_grad"], 0.5 * (r[0]["local_grad"] + r[1]["local_grad"]), rtol=0, atol=1e-7)
    assert not torch.equal(r[0]["local_grad"], r[1]["local_grad"])
    # each rank's local gradient is the reference gradient of ITS shard alone (per-GPU negatives and BN)
    from oracle import dpc_oracle as O
    p = O.make_params_pcg("resnet18", WIDTHS)
    xg = O.make_input_pcg(BPER * world, 8, 5, SIZE)
    for i in range(world):
        ones = [torch.ones(BPER, WIDTHS[3], 2, 2) for _ in range(8)]
        loss, accs, grads, _ = O.train_step_reference(p, xg[i * BPER:(i + 1) * BPER], "resnet18", 3, ones)
        assert abs(r[i]["res"][0].item() - loss.item()) < 1e-4
        for k, g in grads.items():
            o, n = r[i]["offsets"][k]
            mine = r[i]["local_grad"][o:o + n].view(g.shape)
            assert (mine - g).abs().max().item() < 2e-3 * max(g.abs().max().item(), 1e-6), k


# ---------------------------------------------------------------- the drop-in module under DistributedDataParallel
def _ddp_rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpc_amd import _lib as L
    from dpc_amd.model import DPC_RNN
    from dpc_amd.parallel import shard_of
    from oracle import dpc_oracle as O
    model = DPC_RNN(sample_size=SIZE, num_seq=8, seq_len=5, pred_step=3, network="resnet18", widths=WIDTHS,
                    _simulator=L.load_emulator())
    model.load_state_dict(O.make_params_pcg("resnet18", WIDTHS), strict=True)
    model.eval()  # dropout off (BN stays batch-stat); the reference's loop lines follow (dpc/main.py:198-231)
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    x = O.make_input_pcg(BPER * world, 8, 5, SIZE)[shard_of(BPER * world, world, rank)].contiguous()
    score_, mask_ = ddp(x)
    B, NP, SQ, B2, NS, _ = mask_.size()
    target = (mask_ == 1).view(B * NP * SQ, B2 * NS * SQ).to(int).argmax(dim=1)
    loss = torch.nn.CrossEntropyLoss()(score_.view(B * NP * SQ, B2 * NS * SQ), target)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    torch.save({"grads": grads, "loss": loss.detach()}, os.path.join(out_dir, f"ddp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_module_under_ddp(tmp_path):
    """SURVEY section 8b "Callers": DPC_RNN must work under one-process-per-GPU DDP.  DDP averages what the module's
    autograd node hands back: every rank ends with the mean of the per-shard reference gradients."""
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    world, port = 2, 29900 + os.getpid() % 300
    mp.spawn(_ddp_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"ddp{i}.pt") for i in range(world)]
    from oracle import dpc_oracle as O
    p = O.make_params_pcg("resnet18", WIDTHS)
    xg = O.make_input_pcg(BPER * world, 8, 5, SIZE)
    ref = []
    for i in range(world):
        loss, accs, grads, _ = O.train_step_reference(p, xg[i * BPER:(i + 1) * BPER], "resnet18", 3, None)
        assert abs(r[i]["loss"].item() - loss.item()) < 1e-4
        ref.append(grads)
    for k in ref[0]:
        mean = 0.5 * (ref[0][k] + ref[1][k])
        assert torch.equal(r[0]["grads"][k], r[1]["grads"][k]), k
        assert (r[0]["grads"][k] - mean).abs().max().item() < 2e-3 * max(mean.abs().max().item(), 1e-6), k


def test_rccl_defaults_for_data_parallel_runs():
    """few long-lived channels + the same number of CUs left free, only when there is more than one rank; explicit settings win"""
    from dpc_amd.parallel import DEFAULT_CHANNELS, configure_rccl, default_reserve_cus
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    assert configure_rccl(1, env) == {"DPC_RESERVE_CUS": 0} and "NCCL_MAX_NCHANNELS" not in env
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    got = configure_rccl(8, env)
    assert got["NCCL_MAX_NCHANNELS"] == DEFAULT_CHANNELS == int(env["NCCL_MAX_NCHANNELS"]) and "NCCL_MIN_NCHANNELS" not in env
    assert got["DPC_RESERVE_CUS"] == DEFAULT_CHANNELS
    env = {"NCCL_MAX_NCHANNELS": "24", "NCCL_MIN_NCHANNELS": "4"}
    got = configure_rccl(8, env)
    assert env["NCCL_MAX_NCHANNELS"] == "24" and env["NCCL_MIN_NCHANNELS"] == "4" and got["DPC_RESERVE_CUS"] == 24
    assert "HSA_ENABLE_IPC_MODE_LEGACY" in got and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    env = {"DPC_RCCL_CHANNELS": "16", "DPC_RESERVE_CUS": "0", "NCCL_MAX_NCHANNELS": "2"}
    got = configure_rccl(2, env)
    assert env["NCCL_MAX_NCHANNELS"] == "16" == env["NCCL_MIN_NCHANNELS"] and got["DPC_RESERVE_CUS"] == 0
    assert default_reserve_cus(1, {}) == 0 and default_reserve_cus(4, {}) == DEFAULT_CHANNELS


def _clock_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from dpc_amd.parallel import gather_step_times, ranks_agree
    from dpc_amd.main import average_over_ranks
    dev = torch.device("cpu")
    mx, per = gather_step_times(dist, 1.0 + rank, 10, dev, world)          # rank r "took" 1 + r seconds for 10 steps
    ok_all = ranks_agree(dist, True, dev)
    ok_one = ranks_agree(dist, rank != 1, dev)                              # rank 1 "failed its capture": nobody goes on
    vals = average_over_ranks(dist, torch.tensor([1.0 + rank, 10.0 * rank, 0.0, 4.0]), world)
    torch.save({"max": mx, "per": per, "ok_all": ok_all, "ok_one": ok_one, "vals": vals}, os.path.join(out_dir, f"clock{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_clock_and_agreement_two_ranks(tmp_path):
    """the multi-rank lines of bench.py (per-rank times gathered, the MAX is the clock; side schedules only when every rank agrees) and
    dpc_amd.main's averaged metrics, over gloo: their first execution with world > 1 must not be the driver's 8-GPU run"""
    world, port = 2, 29900 + os.getpid() % 90
    mp.spawn(_clock_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"clock{i}.pt") for i in range(world)]
    for x in r:
        assert x["max"] == 2.0 and x["per"] == [100.0, 200.0]
        assert x["ok_all"] is True and x["ok_one"] is False
        assert torch.equal(x["vals"], torch.tensor([1.5, 5.0, 0.0, 4.0]))
    from dpc_amd.parallel import gather_step_times, ranks_agree
    assert gather_step_times(None, 3.0, 6, torch.device("cpu"), 1) == (3.0, [500.0]) and ranks_agree(None, False, None) is False
