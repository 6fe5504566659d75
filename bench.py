#!/usr/bin/env python
"""bench.py -- clips/sec of the DPC-RNN train step on N MI355X (one process per GPU).

A step = forward + CE/top-k + backward + (RCCL gradient all-reduce) + Adam over one batch of
synthetic video already resident in HBM.  N=1 workload = BASELINE.json configs[1]:
resnet18 2d3d, img_dim 128, seq_len 5, num_seq 8, pred_step 3, batch 128 per GPU, bf16 compute
(f32 master weights / accumulation).  For N>1 the driver launches this file under
torch.distributed.run; every rank runs the same per-GPU batch (weak scaling) and all-reduces the
flat f32 gradient (58.3 MB) once per step.

Prints ONE JSON line on rank 0, including
  roofline     -- the dominant kernel (implicit-GEMM conv, MFMA-bound): algorithmic FLOPs per
                  launch / average launch duration, HIP events on the launch stream, timed region only
  cpu_baseline -- the CPU oracle (torch-CPU port of the reference step) on configs[0] (batch 4),
                  rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16 = 2500.0  # TFLOP/s dense, /opt/skills/guides/MI355X_MICROARCH.md:42
MFMA_PEAK_F32 = 157.3    # ibid. :41


def cpu_baseline(seconds_budget=20.0):
    """oracle (kind 'port'): forward+CE+top-k+backward+Adam of configs[0] on the host cores"""
    from oracle import dpc_oracle as O
    torch.manual_seed(0)
    p = O.init_params_reference_style("resnet18", seed=0)
    x = torch.randn(4, 8, 3, 5, 128, 128)
    names = O.unique_param_names(p)
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    masks = [torch.ones(4, 256, 4, 4) for _ in range(8)]

    def step(i):
        _, _, grads, _ = O.train_step_reference(p, x, "resnet18", 3, masks)
        for k in names:
            O.adam_step(p[k], grads[k], m[k], v[k], i + 1)

    step(0)  # warm-up (oneDNN primitive creation)
    t0 = time.time()
    n = 0
    while True:
        step(n + 1)
        n += 1
        if time.time() - t0 > seconds_budget or n >= 20:
            break
    dt = time.time() - t0
    return {"value": round(4 * n / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"configs[0] r18/128/B=4 fwd+loss+bwd+Adam, {n} steps in {dt:.1f}s (host cpu_count={os.cpu_count()})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="clips per GPU")
    ap.add_argument("--net", default="resnet18")
    ap.add_argument("--img_dim", type=int, default=128)
    ap.add_argument("--pred_step", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under a launcher the RCCL path is exercised even with one rank
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dpc_amd.engine import DPCEngine, KernelTimer
    from dpc_amd.model import DPC_RNN
    from dpc_amd.parallel import make_allreduce

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    eng = DPCEngine(args.net, args.img_dim, 8, 5, args.pred_step, args.batch, dev, cdt)
    init = DPC_RNN(args.img_dim, network=args.net, pred_step=args.pred_step, seed=0)  # reference init, same on all ranks
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    del init
    g = torch.Generator(dev).manual_seed(1234 + rank)
    block = torch.randn(args.batch, 8, 3, 5, args.img_dim, args.img_dim, device=dev, generator=g)
    allreduce = make_allreduce(dist, world, force=dist is not None)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = eng.train_step(block, allreduce=allreduce)
    timer = None
    if not args.no_roofline:
        timer = KernelTimer(["dpc_conv_igemm", "dpc_conv_wgrad"])
        eng.timer = timer
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = eng.train_step(block, allreduce=allreduce)
    sync()
    dt = time.perf_counter() - t0
    eng.timer = None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    loss = res.cpu().tolist()

    out = None
    if rank == 0:
        clips = args.batch * world * args.steps
        out = {
            "metric": "clips/sec (train step, B x8x3x5xHxW)", "value": round(clips / dt, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.net} 2d3d, img_dim {args.img_dim}, seq_len 5, num_seq 8, pred_step {args.pred_step}, "
                                   f"batch {args.batch}/GPU, full train step (fwd+CE/top-k+bwd+all-reduce+Adam)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "init": "reference init, random"},
            "final_loss": round(loss[0], 4),
        }
        if timer is not None:
            s = timer.summary()
            ig = s.get("dpc_conv_igemm")
            peak = MFMA_PEAK_BF16 if args.dtype == "bf16" else MFMA_PEAK_F32
            traffic = None
            try:  # HBM bytes per launch from the rocprofv3 PMC passes (scripts/gpu_pmc.sh + scripts/pmc_traffic.py)
                import glob
                cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
                if cands and args.dtype == "bf16" and args.batch == 128 and args.net == "resnet18" and args.img_dim == 128:
                    traffic = round(json.load(open(cands[-1]))["dpc_conv_igemm"]["hbm_bytes_per_launch"] / 1e9, 4)
            except Exception:
                traffic = None
            if ig and ig["ms"] > 0:
                ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
                out["roofline"] = {
                    "kernel": "dpc_conv_igemm (igemm_ws_kernel + conv_halo(_ws)_kernel + igemm_kernel: conv fwd + input-grad + 1x1/score GEMMs)",
                    "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_unit": "GB of HBM traffic per launch (rocprofv3 PMC, profiles/*_pmc_traffic.json)",
                    "algorithmic_GB_per_launch": round(ig["bytes"] / ig["launches"] / 1e9, 4),
                    "launches_per_step": ig["launches"] // args.steps,
                    "avg_launch_us": round(1e3 * ig["ms"] / ig["launches"], 2),
                    "flops_per_launch": round(ig["flops"] / ig["launches"] / 1e9, 3),
                    "flops_unit": "GFLOP (algorithmic)",
                    "ms_per_step": round(ig["ms"] / args.steps, 3),
                    "algorithmic_GBps": round(ig["bytes"] / (ig["ms"] * 1e-3) / 1e9, 1),
                }
            wg = s.get("dpc_conv_wgrad")
            if wg and wg["ms"] > 0:
                out["wgrad_kernel"] = {"ms_per_step": round(wg["ms"] / args.steps, 3),
                                       "achieved_TFLOPs": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2),
                                       "launches_per_step": wg["launches"] // args.steps}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
