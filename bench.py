#!/usr/bin/env python
"""bench.py -- clips/sec of the DPC-RNN train step on N MI355X (one process per GPU).

A step = forward + CE/top-k + backward + (RCCL gradient all-reduce) + Adam over one batch of
synthetic video already resident in HBM.  Default workload = BASELINE.json configs[1] ("cfg2"):
resnet18 2d3d, img_dim 128, seq_len 5, num_seq 8, pred_step 3, batch 128 per GPU, bf16 compute
(f32 master weights / accumulation).  ``--config cfg4`` / ``cfg5`` run the single-GPU shard of
configs[3] / configs[4] (resnet34, 224^2, batch 44 / pred_step 5, batch 64).  For N>1 the driver
launches this file under torch.distributed.run; every rank runs the same per-GPU batch (weak scaling)
and all-reduces the flat f32 gradient (58.3 MB r18 / 131.8 MB r34) once per step.

Timing: ``value`` comes from K steps with NO instrumentation (barrier + synchronize on both sides, max
over ranks).  The roofline objects come from a second, separately instrumented pass (HIP events around
the selected C-ABI launches on the launch stream) that does not touch ``value``.

Prints ONE JSON line on rank 0, including
  roofline     -- the dominant kernel family (implicit-GEMM conv, MFMA-bound): algorithmic FLOPs per
                  launch / average launch duration
  score_gemm   -- the contrastive score contraction (forward + both backward GEMMs), the metric's named
                  kernel: algorithmic FLOPs / time against the dense bf16 MFMA peak
  hbm_family   -- BN / pool / pack kernels (HBM-bound): algorithmic bytes / time against 8 TB/s
  cpu_baseline -- the CPU oracle (torch-CPU port of the reference step) on configs[0] (batch 4),
                  rank 0, N=1 only, best of a thread-count sweep.
  also         -- (default cfg2 run on one GPU only) the 224^2 configurations of BASELINE.json measured in the same process:
                  the single-GPU shards of configs[3] (cfg4) and configs[4] (cfg5), a few hipGraph replays each, with the
                  conv family's roofline fraction.  They never enter `value`.
"""
import hashlib
import os as _os
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # before HIP initialises; see dpc_amd/__init__.py (this file touches the GPU before it imports the package)
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
HBM_PEAK_GBS = 8000.0    # ibid. :35

CONFIGS = {  # BASELINE.json "configs" -> per-GPU shard
    "cfg2": dict(net="resnet18", img_dim=128, pred_step=3, batch=128),
    "cfg4": dict(net="resnet34", img_dim=224, pred_step=3, batch=44),
    "cfg5": dict(net="resnet34", img_dim=224, pred_step=5, batch=64),
}


def csrc_sha16():
    """fingerprint of the kernel sources: a PMC traffic file measured on other kernels is stale (no .git on the GPU box)"""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dpc_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h")):
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def live_pmc_traffic(config="cfg2", timeout_s=90):
    """HBM bytes per launch of the conv family, measured IN THIS RUN: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE -- separate
    --pmc runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of a child `bench.py` that launches two
    eager steps of the same workload, summarised with the guide's gfx950 correction (scripts/pmc_traffic.py; both child steps run the same static schedule, so averaging over them
    changes nothing).  Returns (summary dict, None) or (None, reason).  Costs ~40 s, at most 2 x 90 s when a pass hangs (it is killed, the line
    then falls back to the tracked profile); `--pmc file` reads the tracked profile instead, `--pmc off` skips it."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import pmc_traffic as PT
    finally:
        sys.path.pop(0)
    tmp = tempfile.mkdtemp(prefix="dpc_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
             "--no-also", "--no-graph", "--pmc", "off"]
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            # its own session: on a timeout the whole group goes (rocprofv3 AND the python it started), nothing keeps the GPU busy behind us
            proc = subprocess.Popen(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "--"] + child, cwd="/tmp", env=env,
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                so, se = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.communicate()
                return None, f"the {counter} pass did not finish in {timeout_s} s"
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if proc.returncode != 0 or not dbs:
                return None, f"the {counter} pass failed (rc {proc.returncode}): {(se or so)[-160:]}"
            got[counter] = PT.per_kernel(dbs[0], counter)
        return PT.summarise(got["FETCH_SIZE"], got["WRITE_SIZE"]), None
    except Exception as e:   # reported in the line, never fatal to it
        return None, f"{type(e).__name__}: {str(e)[:160]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def score_block(eng, sc, steps, peak):
    """the `score_gemm` object of a bench line from KernelTimer's tag:score totals of `steps` instrumented steps"""
    R = eng.R
    ach = sc["flops"] / (sc["ms"] * 1e-3) / 1e12
    return {
        "what": f"contrastive score {R}x{R}x{eng.D}: forward + d_pred + d_feature_inf contractions ({eng.score_mode} path)",
        "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
        "us_per_step": round(1e3 * sc["ms"] / steps, 1), "launches_per_step": sc["launches"] // steps,
        "flops_per_step": round(sc["flops"] / steps / 1e9, 2), "flops_unit": "GFLOP (algorithmic: 3 x 2 R^2 D)",
        "score_bytes": R * R * (2 if "bf16 logits" in eng.score_mode else 4), "logits": "bf16" if "bf16 logits" in eng.score_mode else "f32",
        "includes": ("the softmax statistics / loss and the recomputation of dS inside the backward (no [R][R] tensor is "
                     "written)" if eng.score_mode == "fused" else "the three GEMMs with the reductions of their split-K slabs (CE/top-k and dS are a separate kernel)"),
    }


def side_config(name, dev, steps=8, warmup=2, roof_steps=2, score_path="auto"):
    """one of the 224^2 configurations, measured after the headline run in the same process (single GPU, no exchange)"""
    from dpc_amd.engine import DPCEngine, KernelTimer
    from dpc_amd.model import DPC_RNN
    cfg = CONFIGS[name]
    net, img, P, batch = cfg["net"], cfg["img_dim"], cfg["pred_step"], cfg["batch"]
    eng = DPCEngine(net, img, 8, 5, P, batch, dev, torch.bfloat16, seed=233, score_path=score_path)
    init = DPC_RNN(img, network=net, pred_step=P, seed=0)
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    del init
    block = torch.randn(batch, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(1234))
    step = eng.capture_train_step(block)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": f"{net} 2d3d, img_dim {img}, pred_step {P}, batch {batch}/GPU, bf16, full train step, hipGraph replay",
           "value": round(batch * steps / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "final_loss": round(res.cpu().tolist()[0], 4), "score_path": eng.score_mode}
    timer = KernelTimer(["dpc_conv_igemm", "dpc_conv_igemm_ex", "dpc_score_fwd", "dpc_score_bwd", "dpc_gemm_nt_splitk", "dpc_gemm_tn_splitk",
                         "dpc_reduce_unpack"])
    eng.timer = timer
    for _ in range(roof_steps):
        eng.train_step(block)
    torch.cuda.synchronize()
    eng.timer = None
    summ = timer.summary(roof_steps)
    ig = summ.get("dpc_conv_igemm")
    if ig and ig["ms"] > 0:
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": "dpc_conv_igemm", "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_BF16,
                           "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_BF16, 4), "ms_per_step": round(ig["ms"] / roof_steps, 3),
                           "executed_frac": round(ig["executed_flops"] / (ig["ms"] * 1e-3) / 1e12 / MFMA_PEAK_BF16, 4)}
    sc = summ.get("tag:score")
    if sc and sc["ms"] > 0:
        out["score_gemm"] = score_block(eng, sc, roof_steps, MFMA_PEAK_BF16)
    del eng, step, block
    torch.cuda.empty_cache()
    return out


def f32_mode(dev, steps=4, matmul="exact"):
    """cfg2 in the f32 parity mode (the mode north_star's 1e-3 tolerance is stated in): a few hipGraph replays, same process.
    matmul="bf16x6": the same f32 tensors, contractions on the bf16 matrix pipe with three-way split operands (dpc_set_f32_matmul)"""
    from dpc_amd.engine import DPCEngine
    from dpc_amd.model import DPC_RNN
    cfg = CONFIGS["cfg2"]
    net, img, P, batch = cfg["net"], cfg["img_dim"], cfg["pred_step"], cfg["batch"]
    eng = DPCEngine(net, img, 8, 5, P, batch, dev, torch.float32, seed=233, f32_matmul=matmul)
    init = DPC_RNN(img, network=net, pred_step=P, seed=0)
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    del init
    block = torch.randn(batch, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(1234))
    step = eng.capture_train_step(block, warmup=1)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    what = ("exact f32 MFMA chains" if matmul == "exact" else
            "f32 tensors, contractions as bf16x6: operands split three ways onto the bf16 matrix pipe, f32 accumulate")
    out = {"workload": f"cfg2 in f32 end to end ({what}): score / mask / top-k match the reference's CPU path within 1e-3 in this mode "
                       "(tests/test_engine_gpu.py)", "dtype": "f32" if matmul == "exact" else "f32 (bf16x6 matmul)", "value": round(batch * steps / dt, 2), "unit": "clips/s",
           "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "final_loss": round(res.cpu().tolist()[0], 4)}
    del eng, step, block
    torch.cuda.empty_cache()
    return out


def one_stream(dev, steps=20, rounds=3):
    """cfg2 with the weight gradients on the main stream (DPC_WGRAD_STREAM=0) against the two-stream schedule of the headline line,
    PAIRED: both engines live in this process, both steps are captured, and blocks of `steps` replays alternate (two, one, two, one,
    ...), so the two schedules see the same clocks and the same allocator state.  Round 4 compared this leg -- run minutes into the
    process -- with the headline measured in the process's first second and read 1.0035; an alternating fresh-process A/B of the same
    two schedules on one box says 1.033 / 1.030 / 1.029 at cfg2 / cfg4 / cfg5 (profiles/r05_two_stream_ab.txt)."""
    from dpc_amd.engine import DPCEngine
    from dpc_amd.model import DPC_RNN
    cfg = CONFIGS["cfg2"]
    net, img, P, batch = cfg["net"], cfg["img_dim"], cfg["pred_step"], cfg["batch"]
    init = DPC_RNN(img, network=net, pred_step=P, seed=0)
    params = {k: v.detach() for k, v in init.named_parameters()}
    block = torch.randn(batch, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(1234))

    def build(streams):
        prev = os.environ.get("DPC_WGRAD_STREAM")
        os.environ["DPC_WGRAD_STREAM"] = str(streams)
        try:
            eng = DPCEngine(net, img, 8, 5, P, batch, dev, torch.bfloat16, seed=233)
        finally:
            if prev is None:
                os.environ.pop("DPC_WGRAD_STREAM", None)
            else:
                os.environ["DPC_WGRAD_STREAM"] = prev
        eng.load_params(params)
        return eng, eng.capture_train_step(block)

    def timed(step):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    e2, s2 = build(1)
    e1, s1 = build(0)
    assert e2._side is not None and e1._side is None
    for _ in range(3):
        s2(); s1()
    two, one = [], []
    for _ in range(rounds):
        two.append(timed(s2))
        one.append(timed(s1))
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    out = {"workload": "cfg2, bf16, hipGraph replay, ONE stream (no weight gradients beside the next unit's BatchNorm backward) against the "
                       f"two-stream schedule, paired in this process: {rounds} alternating blocks of {steps} steps each",
           "value": round(batch / (med(one) * 1e-3), 2), "unit": "clips/s", "ms_per_step": round(med(one), 3), "steps": steps * rounds,
           "two_stream_ms_per_step": round(med(two), 3), "all_ms_per_step": {"two": [round(v, 3) for v in two], "one": [round(v, 3) for v in one]},
           "two_stream_speedup": round(med(one) / med(two), 4)}
    del e1, e2, s1, s2, block
    torch.cuda.empty_cache()
    return out


def module_loop(dev, steps=10, warmup=3):
    """the reference's own loop lines (dpc/main.py:198-231: model(x) -> CrossEntropyLoss -> zero_grad / backward / optimizer.step)
    over the drop-in module dpc_amd.model.DPC_RNN at cfg2, bf16 compute: what a user who only swaps the import gets"""
    from dpc_amd.model import DPC_RNN
    from dpc_amd.optim import Adam
    cfg = CONFIGS["cfg2"]
    net, img, P, batch = cfg["net"], cfg["img_dim"], cfg["pred_step"], cfg["batch"]
    model = DPC_RNN(img, num_seq=8, seq_len=5, pred_step=P, network=net, seed=0).to(dev).bfloat16()
    model.train()
    optimizer = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    criterion = torch.nn.CrossEntropyLoss()
    x = torch.randn(batch, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(1234))
    target = [None]

    def one():
        score_, mask_ = model(x)
        B, NP, SQ, B2, NS, _ = mask_.size()
        if target[0] is None:   # the reference builds the target once (dpc/main.py:211, process_output)
            target[0] = (mask_ == 1).view(B * NP * SQ, B2 * NS * SQ).to(int).argmax(dim=1)
        loss = criterion(score_.view(B * NP * SQ, B2 * NS * SQ), target[0])
        del score_
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        return loss

    for _ in range(warmup):
        loss = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "cfg2 through dpc_amd.model.DPC_RNN(...).bfloat16() + torch CrossEntropyLoss + dpc_amd.optim.Adam: the reference's "
                       "loop lines, launched kernel by kernel (no hipGraph), score materialised and returned",
           "value": round(batch * steps / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "final_loss": round(float(loss.item()), 4)}
    del model, optimizer, x
    torch.cuda.empty_cache()
    return out


def cpu_baseline(seconds_budget=24.0):
    """oracle (kind 'port'): forward+CE+top-k+backward+Adam of configs[0] on the host cores; the thread count is
    swept (B=4 oversubscribes a 128-thread pool) and the best rate is reported with the count that gave it."""
    from oracle import dpc_oracle as O
    torch.manual_seed(0)
    p = O.init_params_reference_style("resnet18", seed=0)
    x = torch.randn(4, 8, 3, 5, 128, 128)
    names = O.unique_param_names(p)
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    masks = [torch.ones(4, 256, 4, 4) for _ in range(8)]
    it = [0]

    def step():
        it[0] += 1
        _, _, grads, _ = O.train_step_reference(p, x, "resnet18", 3, masks)
        for k in names:
            O.adam_step(p[k], grads[k], m[k], v[k], it[0])

    ncpu = os.cpu_count() or 8
    default_threads = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, 64, default_threads) if t <= max(ncpu, 8)})
    per = seconds_budget / len(cands)
    sweep, best = {}, None
    t_all = time.time()
    for t in cands:
        torch.set_num_threads(t)
        step()  # warm-up (oneDNN primitive creation for this thread count)
        t0 = time.time()
        n = 0
        while True:
            step()
            n += 1
            if time.time() - t0 > per * 0.6 or n >= 6:
                break
        rate = 4 * n / (time.time() - t0)
        sweep[str(t)] = round(rate, 3)
        if best is None or rate > best[0]:
            best = (rate, t, n)
    torch.set_num_threads(default_threads)
    return {"value": round(best[0], 3), "unit": "clips/s", "cores": best[1], "kind": "port",
            "sample": f"configs[0] r18/128/B=4 fwd+loss+bwd+Adam; thread sweep {sweep} clips/s, best at {best[1]} threads "
                      f"({best[2]} timed steps), {time.time() - t_all:.0f}s total (host cpu_count={ncpu})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json configuration (per-GPU shard)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (overrides --config)")
    ap.add_argument("--net", default=None)
    ap.add_argument("--img_dim", type=int, default=None)
    ap.add_argument("--pred_step", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x6"],
                    help="f32 = the parity mode (1e-3 vs the reference); bf16x6 = f32 tensors with the contractions on the bf16 matrix pipe (parity-grade)")
    ap.add_argument("--roofline-steps", type=int, default=4, help="steps of the separately instrumented pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the 224^2 side measurements (cfg4 / cfg5) of the default run")
    ap.add_argument("--schedules", action="store_true",
                    help="multi-rank runs: after the timed region also time the other gradient-exchange schedules (two more captures with other "
                         "collectives; opt-in: a side measurement must not be able to cost a scaling run its line)")
    ap.add_argument("--no-schedules", action="store_true", help="(accepted for compatibility: the side schedules are opt-in since round 5)")
    ap.add_argument("--no-graph", action="store_true", help="launch the step kernel by kernel instead of replaying a hipGraph")
    ap.add_argument("--fresh-input", action="store_true", help="draw a new N(0,1) block on the device inside every timed step (torch's randn kernel, "
                    "captured into the graph): +0.3 ms per step at cfg2.  Default: ONE resident synthetic block, re-used by every step "
                    "(config.input says which)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "live", "file", "off"],
                    help="roofline.traffic: live = two rocprofv3 --pmc passes of a child run (default at cfg2 / bf16 / one GPU: 'auto' tries it and "
                         "falls back to the tracked profiles/*_pmc_traffic.json, flagged); file = the tracked profile only; off = null")
    ap.add_argument("--score-path", default="auto", choices=["auto", "fused", "materialised"],
                    help="contrastive score + loss: fused (no [R][R] tensor in HBM) or materialised; auto = materialised (the faster one at cfg2 and at cfg5)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("batch", "net", "img_dim", "pred_step"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    net, img, P, batch = cfg["net"], cfg["img_dim"], cfg["pred_step"], cfg["batch"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    rccl_knobs = {}
    under_launcher = world > 1 or "RANK" in os.environ   # under a launcher the RCCL path is exercised even with one rank
    if under_launcher:
        from dpc_amd.parallel import configure_rccl
        rccl_knobs = configure_rccl(world)  # before the first HIP call: channel count of the gradient all-reduce, CU carve-out (parallel.py)
    # DPC_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo carrying the collectives -- the multi-rank control flow of this file (the
    # three-graph replay with a live exchange between the graphs, barriers, the clock's all_gather, the side schedules and their
    # rank agreement) executed on a ONE-GPU box (scripts/gpu_r6_tworank.sh).  RCCL refuses two ranks on one device; the line such a
    # run prints says "rehearsal" in `scaling` and is not a measurement.
    rehearsal = os.environ.get("DPC_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if under_launcher:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dpc_amd.engine import DPCEngine, KernelTimer, HBM_FAMILY
    from dpc_amd.model import DPC_RNN
    from dpc_amd.parallel import make_allreduce

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    from dpc_amd.parallel import default_reserve_cus
    eng = DPCEngine(net, img, 8, 5, P, batch, dev, cdt, seed=233 + rank, score_path=args.score_path,
                    reserve_cus=default_reserve_cus(world), f32_matmul="bf16x6" if args.dtype == "bf16x6" else "exact")
    init = DPC_RNN(img, network=net, pred_step=P, seed=0)  # reference init, same on all ranks
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    del init
    g = torch.Generator(dev).manual_seed(1234 + rank)
    block = torch.randn(batch, 8, 3, 5, img, img, device=dev, generator=g)
    allreduce = make_allreduce(dist, world, force=dist is not None)
    use_graph, graph_note = not args.no_graph, ""
    step_fn = None
    if use_graph:
        try:
            step_fn = eng.capture_train_step(block, allreduce=allreduce, refill=(lambda: block.normal_()) if args.fresh_input else None)
        except Exception as e:  # report it, never hide it: the line says which launch mode was timed
            use_graph, graph_note = False, f" (hipGraph capture failed: {type(e).__name__}: {str(e)[:120]})"
            torch.cuda.synchronize()
    if step_fn is None:
        step_fn = (lambda: eng.train_step(block.normal_(), allreduce=allreduce)) if args.fresh_input else (lambda: eng.train_step(block, allreduce=allreduce))  # noqa: E731

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step_fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step_fn()
    sync()
    dt = time.perf_counter() - t0
    from dpc_amd.parallel import gather_step_times, ranks_agree
    dt, rank_ms = gather_step_times(dist, dt, args.steps, dev, world)   # max over ranks; per-rank spread (tests/test_parallel_gloo.py)
    loss = res.cpu().tolist()

    # ---- the other exchange schedules, a few steps each (never part of `value`): the default overlaps the tail all-reduce with
    # layer1 + stem backward and leaves reserve_cus CUs to RCCL; "overlap_all_cus" = the same with every CU claimed by the
    # persistent grids, "serial" = one all-reduce of the whole arena after the backward pass
    schedules = None
    if dist is not None and use_graph and allreduce is not None and args.schedules and not args.no_schedules:
        def agree(ok: bool) -> bool:
            return ranks_agree(dist, ok, dev)

        failed = []   # once any rank failed a side schedule, every rank skips the rest: their collectives would no longer pair up

        def side(exch, reserve, n=max(4, min(args.steps, 10))):
            """a side measurement must never cost the line.  A failure on ONE rank (capture refused, out of memory) must not leave the
            other ranks inside collectives nobody answers: the ranks agree after the capture and again after the timed loop, and
            whatever one rank could not do, no rank goes on with.  What this cannot cure is a rank dying INSIDE a
            gradient exchange; RCCL's own watchdog ends that run."""
            if failed:
                return f"skipped: {failed[0]}"
            keep, note, fn, t = eng.reserve_cus, None, None, None
            eng.reserve_cus = reserve
            try:
                try:
                    fn = eng.capture_train_step(block, allreduce=exch)
                except Exception as e:
                    note = f"failed: {type(e).__name__}: {str(e)[:120]}"
                    torch.cuda.synchronize()
                if not agree(fn is not None):
                    failed.append(note or "capture failed on another rank")
                    return failed[0]
                try:
                    fn(); fn()
                    sync()
                    t1 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    sync()
                    t = (time.perf_counter() - t1) / n
                except Exception as e:
                    note = f"failed: {type(e).__name__}: {str(e)[:120]}"
                    torch.cuda.synchronize()
                if not agree(t is not None):
                    failed.append(note or "timed loop failed on another rank")
                    return failed[0]
                tt = torch.tensor([t], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return round(1e3 * tt.item(), 3)
            finally:
                eng.reserve_cus = keep
        schedules = {"default": {"ms_per_step": round(1e3 * dt / args.steps, 3), "reserve_cus": eng.reserve_cus, "exchange": "two buckets, overlapped"}}
        if eng.reserve_cus:
            schedules["overlap_all_cus"] = {"ms_per_step": side(allreduce, 0), "reserve_cus": 0}
        schedules["serial"] = {"ms_per_step": side(lambda flat: allreduce(flat), 0), "reserve_cus": 0, "exchange": "one all-reduce after the backward pass"}

    # ---- separately instrumented pass (never part of `value`): HIP events around the selected launches
    timer = None
    if not args.no_roofline and rank == 0:
        timer = KernelTimer(["dpc_conv_igemm", "dpc_conv_igemm_ex", "dpc_conv_wgrad", "dpc_score_fwd", "dpc_score_bwd", "dpc_gemm_nt_splitk", "dpc_gemm_tn_splitk",
                             "dpc_reduce_unpack"] + list(HBM_FAMILY))
        eng.timer = timer
        for _ in range(args.roofline_steps):
            eng.train_step(block, allreduce=None)
        torch.cuda.synchronize()
        eng.timer = None
    rs = max(args.roofline_steps, 1)

    out = None
    if rank == 0:
        clips = batch * world * args.steps
        out = {
            "metric": "clips/sec (train step, B x8x3x5xHxW)", "value": round(clips / dt, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "rehearsal (ranks share cuda:0, gloo): not a measurement" if rehearsal else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.config}: {net} 2d3d, img_dim {img}, seq_len 5, num_seq 8, pred_step {P}, "
                                   f"batch {batch}/GPU, full train step (fwd+CE/top-k+bwd+all-reduce+Adam)",
                       "global_batch": batch * world, "parallelism": f"dp{world}", "init": "reference init, random",
                       "input": ("a new N(0,1) block drawn on the device inside every timed step" if args.fresh_input else
                                 "one resident N(0,1) block [B,8,3,5,H,W] f32 drawn on the device before the timed region and re-used by every step "
                                 "(--fresh-input draws one per step inside the graph)"),
                       "launch": "hipGraph replay" if use_graph else "kernel by kernel" + graph_note,
                       "streams": 2 if getattr(eng, "_side", None) is not None else 1},   # weight gradients beside the next BatchNorm backward
            "final_loss": round(loss[0], 4),
        }
        if dist is not None:
            out["per_rank_ms_per_step"] = {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3), "all": [round(v, 3) for v in rank_ms]}
            out["config"]["rccl_knobs"] = rccl_knobs
            if schedules:
                out["schedules"] = schedules
        if timer is not None:
            s = timer.summary(rs)
            ig = s.get("dpc_conv_igemm")
            peak = MFMA_PEAK_BF16 if args.dtype == "bf16" else MFMA_PEAK_F32
            traffic, traffic_src = None, None
            std = args.dtype == "bf16" and args.config == "cfg2" and batch == 128 and args.net is None and args.img_dim is None and args.pred_step is None
            if args.pmc in ("auto", "live") and std and world == 1 and dist is None:
                summ, why = live_pmc_traffic(args.config)
                if summ is not None and summ["dpc_conv_igemm"]["launches"] > 0:
                    traffic = round(summ["dpc_conv_igemm"]["hbm_bytes_per_launch"] / 1e9, 4)
                    traffic_src = {"live": True, "how": "two rocprofv3 passes in this run (--kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate), a child "
                                                        "bench.py launching 2 eager steps of this workload; FETCH_SIZE x2 (gfx950 wide-load correction)",
                                   "launches_counted": summ["dpc_conv_igemm"]["launches"], "csrc_sha16": csrc_sha16(), "stale": False}
                else:
                    traffic_src = {"live": False, "live_failed": why}
            if traffic is None and args.pmc in ("auto", "file"):
                try:  # fallback: the tracked profile of an earlier session (scripts/gpu_pmc_traffic.sh); it says which kernel sources it saw
                    import glob
                    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
                    if cands and std:
                        tj = json.load(open(cands[-1]))
                        traffic = round(tj["dpc_conv_igemm"]["hbm_bytes_per_launch"] / 1e9, 4)
                        traffic_src = dict(traffic_src or {}, live=False, file=os.path.basename(cands[-1]), measured_on_csrc_sha16=tj.get("csrc_sha16"),
                                           measured_on_git_head=tj.get("git_head"), this_build_csrc_sha16=csrc_sha16(),
                                           stale=tj.get("csrc_sha16") != csrc_sha16())
                except Exception:
                    traffic = None
            if ig and ig["ms"] > 0:
                ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
                out["roofline"] = {
                    "kernel": "dpc_conv_igemm (igemm_ws_kernel + conv_halo(_ws)_kernel + igemm_kernel: conv fwd + input-grad + 1x1 GEMMs)",
                    "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_unit": "GB of HBM traffic per launch (rocprofv3 PMC: see traffic_source)",
                    "traffic_source": traffic_src,
                    "executed": {"achieved": round(ig["executed_flops"] / (ig["ms"] * 1e-3) / 1e12, 2), "frac": round(ig["executed_flops"] / (ig["ms"] * 1e-3) / 1e12 / peak, 4),
                                 "what": "the same launches charged only with the MACs the matrix pipe runs: igemm_ws skips the temporal taps of a 3x3x3 conv "
                                         "that read zero padding (7 of 9 at T = 3, 4 of 6 at T = 2); `achieved` / `frac` above are the algorithmic rate"},
                    "algorithmic_GB_per_launch": round(ig["bytes"] / ig["launches"] / 1e9, 4),
                    "launches_per_step": ig["launches"] // rs,
                    "avg_launch_us": round(1e3 * ig["ms"] / ig["launches"], 2),
                    "flops_per_launch": round(ig["flops"] / ig["launches"] / 1e9, 3),
                    "flops_unit": "GFLOP (algorithmic)",
                    "ms_per_step": round(ig["ms"] / rs, 3),
                    "algorithmic_GBps": round(ig["bytes"] / (ig["ms"] * 1e-3) / 1e9, 1),
                    "timed": f"separate instrumented pass of {rs} steps (one stream, HIP events on the launch stream; per launch position the median over the steps)",
                }
            wg = s.get("dpc_conv_wgrad")
            if wg and wg["ms"] > 0:
                out["wgrad_kernel"] = {"ms_per_step": round(wg["ms"] / rs, 3),
                                       "achieved_TFLOPs": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2),
                                       "launches_per_step": wg["launches"] // rs}
            sc = s.get("tag:score")
            if sc and sc["ms"] > 0:
                out["score_gemm"] = score_block(eng, sc, rs, peak)
            hb = [s[n] for n in HBM_FAMILY if n in s]
            if hb:
                ms = sum(d["ms"] for d in hb)
                by = sum(d["bytes"] for d in hb)
                out["hbm_family"] = {
                    "kernels": "bn_apply / bn_bwd_reduce / bn_bwd_apply / stem pool fwd+bwd / input pack",
                    "bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_step": round(ms / rs, 3),
                    "algorithmic_GB_per_step": round(by / rs / 1e9, 3), "launches_per_step": sum(d["launches"] for d in hb) // rs,
                }
        if world == 1 and dist is None and args.config == "cfg2" and args.batch is None and args.dtype == "bf16" and not args.no_also:
            del eng, step_fn, block
            torch.cuda.empty_cache()
            out["also"] = {}
            for name, fn in (("cfg4", lambda: side_config("cfg4", dev)), ("cfg5", lambda: side_config("cfg5", dev)),
                             ("cfg5_fused_score", lambda: side_config("cfg5", dev, steps=6, score_path="fused")),
                             ("one_stream", lambda: one_stream(dev)), ("module", lambda: module_loop(dev)), ("f32", lambda: f32_mode(dev)),
                             ("f32_bf16x6", lambda: f32_mode(dev, matmul="bf16x6"))):
                try:
                    out["also"][name] = fn()
                except Exception as e:  # reported, never hidden
                    out["also"][name] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
                    torch.cuda.synchronize()
            if "value" in out["also"].get("module", {}):
                out["also"]["module"]["vs_engine_path"] = round(out["also"]["module"]["value"] / out["value"], 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
