"""Diagnosis (run under torch.distributed.run with 1+ ranks): the same 4 train steps
  A eager, no exchange | B eager + two-bucket RCCL exchange | C hipGraph (3 graphs) + exchange | D hipGraph, no exchange
must leave bit-identical parameters (world 1: the exchange is an identity)."""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from dpc_amd.engine import DPCEngine
from dpc_amd.parallel import make_allreduce
from oracle import dpc_oracle as O

def eng():
    e = DPCEngine("resnet18", 64, 8, 5, 3, 4, dev, torch.bfloat16, seed=233)
    e.load_params(O.init_params_reference_style("resnet18", seed=0))
    return e

x = torch.randn(4, 8, 3, 5, 64, 64, device=dev, generator=torch.Generator(dev).manual_seed(1))
out = {}
for tag in "ABCD":
    e = eng()
    ar = make_allreduce(dist, world, force=True) if tag in "BC" else None
    losses = []
    if tag in "AB":
        for _ in range(4):
            losses.append(e.train_step(x, allreduce=ar)[0].item())
    else:
        rp = e.capture_train_step(x, allreduce=ar, warmup=2)
        for _ in range(2):
            losses.append(rp()[0].item())
    torch.cuda.synchronize()
    out[tag] = (e.flat_p.double().sum().item(), e.flat_p.clone(), losses, e.step_count)
    print(tag, "steps", e.step_count, "param sum %.9f" % out[tag][0], "losses", [round(v, 5) for v in losses], flush=True)
for tag in "BCD":
    print(tag, "== A:", torch.equal(out[tag][1], out["A"][1]), "max |dp|", (out[tag][1] - out["A"][1]).abs().max().item())
dist.barrier()
dist.destroy_process_group()
