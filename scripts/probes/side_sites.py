"""Which weight gradients earn their place on the side stream?  (round 6)

The backward pass forks every weight gradient onto a second stream beside the next unit's BatchNorm backward and joins before
the next input-gradient (engine.side()).  profiles/r05_r18_128_step_timeline.txt shows pairs that lose: in layer1 the HBM-heavy
wgrad_patch<32> beside bn_bwd_apply takes 655-740 us against 307 + 316 alone, and the strided wgrad2 launches of layer3.0 / layer2.0
stretch from 330 / 342 us to 1008 / 879 with the main stream waiting for them.  This probe measures the step with every site moved
back to the main stream one at a time (DPCEngine.side_off), then the union of the sites that gained and a greedy refinement, all in
ONE process on ONE box (captured graph replays, interleaved with the default schedule as the noise reference).

    python scripts/probes/side_sites.py [cfg2|cfg4|cfg5] [steps]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dpc_amd.engine import DPCEngine  # noqa: E402
from dpc_amd.model import DPC_RNN  # noqa: E402

CFG = {"cfg2": ("resnet18", 128, 3, 128), "cfg4": ("resnet34", 224, 3, 44), "cfg5": ("resnet34", 224, 5, 64)}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    net, img, P, B = CFG[cfg]
    dev = torch.device("cuda:0")
    eng = DPCEngine(net, img, 8, 5, P, B, dev, torch.bfloat16)
    init = DPC_RNN(img, network=net, pred_step=P, seed=0)
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    del init
    block = torch.randn(B, 8, 3, 5, img, img, device=dev, generator=torch.Generator(dev).manual_seed(1234))
    sites = ["pack", "head"] + [b.site + u for b in eng.blocks for u in ("c2", "c1")]

    def measure(off, n=steps):
        eng.side_off = frozenset(off)
        fn = eng.capture_train_step(block, warmup=1)
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    out = {"config": cfg, "steps": steps, "sites": sites}
    base = [measure(())]
    singles = {}
    for i, s in enumerate(sites):
        singles[s] = measure((s,))
        if i % 6 == 5:
            base.append(measure(()))
    base.append(measure(()))
    b0 = sum(base) / len(base)
    out["default_ms"] = [round(b, 3) for b in base]
    out["single_site_on_main_ms"] = {s: round(v, 3) for s, v in singles.items()}
    out["single_site_gain_ms"] = {s: round(b0 - v, 3) for s, v in singles.items()}
    noise = max(base) - min(base)
    winners = [s for s in sites if b0 - singles[s] > max(0.02, noise)]
    out["noise_ms"] = round(noise, 3)
    out["winners"] = winners
    out["all_off_ms"] = round(measure(sites), 3)
    cur, cur_ms = list(winners), measure(winners)
    out["union_of_winners_ms"] = round(cur_ms, 3)
    # greedy refinement: add / drop one site at a time while it gains more than the noise
    improved, rounds = True, 0
    while improved and rounds < 3:
        improved, rounds = False, rounds + 1
        for s in sites:
            trial = [x for x in cur if x != s] if s in cur else cur + [s]
            t = measure(trial)
            if cur_ms - t > max(0.02, noise):
                cur, cur_ms, improved = trial, t, True
    out["greedy_off"] = cur
    out["greedy_ms"] = round(cur_ms, 3)
    out["default_again_ms"] = round(measure(()), 3)
    out["greedy_again_ms"] = round(measure(cur), 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
