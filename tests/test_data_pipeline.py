"""SURVEY.md section 8 f4 -- GPU-side input pipeline.  Expectation: the reference's transforms restated with the libraries
they use -- PIL crop / transpose / channel splitting (utils/augmentation.py:99-251) and torchvision's documented ToTensor /
Normalize arithmetic (uint8 HWC -> float CHW / 255; (t - mean) / std in f32) -- then dataset_3d.py:107-111's stack / view /
transpose.  Bit-exact (f32): the kernel does the same f32 operations in the same order.  CPU tier = host simulator."""
import os
import subprocess

import numpy as np
import pytest
import torch
from PIL import Image

from dpc_amd import _lib as L
from dpc_amd.data import MEAN, STD, draw_clip_params, frames_to_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_clip(frames_u8, start, x1, y1, flip, gray, num_seq, seq_len, ds, size, crop=None):
    """dataset_3d.py:94-111 for one video given the loader's draws.  flip 1: k400 recipe (crop, then flip); flip 2: ucf101 recipe
    (flip the frame, crop `crop`, Scale to `size` with the default NEAREST), dpc/main.py:114-132"""
    crop = crop or size
    idx = (np.arange(num_seq)[:, None] * ds * seq_len + start + np.arange(seq_len)[None, :] * ds).reshape(-1)  # idx_sampler
    seq = [Image.fromarray(frames_u8[i]) for i in idx]
    if flip == 2:
        seq = [im.transpose(Image.FLIP_LEFT_RIGHT) for im in seq]                     # RandomHorizontalFlip first
    seq = [im.crop((x1, y1, x1 + crop, y1 + crop)) for im in seq]                     # RandomCrop (consistent box)
    if crop != size:
        seq = [im.resize((size, size), Image.NEAREST) for im in seq]                  # Scale(size=(size, size)), default interpolation
    if flip == 1:
        seq = [im.transpose(Image.FLIP_LEFT_RIGHT) for im in seq]                     # RandomHorizontalFlip
    out = []
    for im, g in zip(seq, gray):
        if g >= 0:                                                                    # RandomGray.grayscale
            ch = np.array(im)[:, :, g]
            im = Image.fromarray(np.dstack([ch, ch, ch]), "RGB")
        t = torch.from_numpy(np.array(im)).permute(2, 0, 1).float().div(255)          # ToTensor
        t = (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)  # Normalize
        out.append(t)
    t_seq = torch.stack(out, 0)
    C_, H, W = out[0].shape
    return t_seq.view(num_seq, seq_len, C_, H, W).transpose(1, 2).contiguous()       # [N, C, SL, H, W]


def run_case(lib, dev, B, F, H0, W0, N, SL, ds, size, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (B, F, H0, W0, 3), dtype=np.uint8)
    aug, gray = draw_clip_params(rng, B, F, N, SL, ds, H0, W0, size)
    aug[0, 3], aug[-1, 3] = 1, 0  # both flip states
    exp = torch.stack([reference_clip(frames[b], *aug[b], gray[b], N, SL, ds, size) for b in range(B)])
    fr, au, gr = (torch.from_numpy(a).to(dev) for a in (frames, aug, gray))
    block = torch.empty(B, N, 3, SL, size, size, device=dev)
    s2d = torch.empty(B * N, SL, size // 2, size // 2, 16, device=dev)
    frames_to_input(lib, fr, au, gr, N, SL, ds, size, block, s2d)
    assert torch.equal(block.cpu(), exp)
    chk = torch.empty_like(s2d)
    lib.call("dpc_pack_input_s2d", block, chk, L.F32, B * N, SL, size, size, lib.stream())
    assert torch.equal(s2d.cpu(), chk.cpu())
    s2b = torch.empty(B * N, SL, size // 2, size // 2, 16, device=dev, dtype=torch.bfloat16)
    frames_to_input(lib, fr, au, None, N, SL, ds, size, None, s2b)  # no RandomGray, bf16 operand only
    exp2 = torch.stack([reference_clip(frames[b], *aug[b], -np.ones(N * SL, np.int8), N, SL, ds, size) for b in range(B)]).to(dev)
    chk2 = torch.empty_like(s2b)
    lib.call("dpc_pack_input_s2d", exp2.contiguous(), chk2, L.BF16, B * N, SL, size, size, lib.stream())
    assert torch.equal(s2b.cpu(), chk2.cpu())


def run_ucf_case(lib, dev, B, F, H0, W0, N, SL, ds, crop, size, seed):
    """the ucf101 recipe: flip (before the crop) -> RandomCrop(crop) -> Scale(size) NEAREST -> RandomGray -> ToTensor -> Normalize"""
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (B, F, H0, W0, 3), dtype=np.uint8)
    aug, gray = draw_clip_params(rng, B, F, N, SL, ds, H0, W0, crop, flip_code=2)
    aug[0, 3], aug[-1, 3] = 2, 0
    exp = torch.stack([reference_clip(frames[b], *aug[b], gray[b], N, SL, ds, size, crop) for b in range(B)])
    fr, au, gr = (torch.from_numpy(a).to(dev) for a in (frames, aug, gray))
    block = torch.empty(B, N, 3, SL, size, size, device=dev)
    frames_to_input(lib, fr, au, gr, N, SL, ds, size, block, None, crop=crop)
    assert torch.equal(block.cpu(), exp)


def test_frames_to_input_emu():
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    run_case(L.load_emulator(), "cpu", 2, 40, 20, 24, 3, 2, 3, 16, seed=1)
    run_ucf_case(L.load_emulator(), "cpu", 2, 40, 40, 50, 3, 2, 3, 28, 16, seed=2)   # crop 28 -> 16 (the 224 -> 128 ratio)
    run_ucf_case(L.load_emulator(), "cpu", 2, 40, 40, 50, 3, 2, 3, 34, 14, seed=3)   # a ratio whose rounding is not symmetric


@pytest.mark.gpu
def test_frames_to_input_gpu():
    run_case(L.load_hip(), "cuda:0", 3, 130, 150, 200, 8, 5, 3, 128, seed=2)  # kinetics-style: short side 150, crop 128
    run_ucf_case(L.load_hip(), "cuda:0", 2, 130, 256, 340, 8, 5, 3, 224, 128, seed=4)  # ucf101: short side 256, crop 224, scale 128


@pytest.mark.gpu
def test_engine_consumes_frames_directly():
    """load_frames() fills the stem operand from uint8 frames; the step that follows equals the step on the f32 block"""
    from dpc_amd.engine import DPCEngine
    from oracle import dpc_oracle as O
    dev, B = "cuda:0", 2
    rng = np.random.default_rng(5)
    frames = torch.from_numpy(rng.integers(0, 256, (B, 130, 80, 96, 3), dtype=np.uint8)).to(dev)
    aug, gray = draw_clip_params(rng, B, 130, 8, 5, 3, 80, 96, 64)
    au, gr = torch.from_numpy(aug).to(dev), torch.from_numpy(gray).to(dev)
    eng = DPCEngine("resnet18", 64, 8, 5, 3, B, dev, torch.float32)
    eng.load_params(O.make_params_pcg("resnet18"))
    block = torch.empty(B, 8, 3, 5, 64, 64, device=dev)
    frames_to_input(eng.lib, frames, au, gr, 8, 5, 3, 64, block, None)
    s_block = eng.forward(block, train=False).clone()
    eng.x_s2d.zero_()
    eng.load_frames(frames, au, gr, ds=3)
    s_frames = eng.forward(None, train=False)
    assert torch.equal(s_block, s_frames)


# ---- the full training recipes, pinned to the REFERENCE'S OWN classes (tests/golden/aug.npz, make_aug_golden.py) ---------------
def _golden_recipe_case(lib, dev, golden_dir, name, seed):
    """random.seed(s); np.random.seed(s); the reference's Compose([...]) of dpc/main.py:114-132 produced the stored clip.  The
    same seeds through dpc_amd.data.draw_* (same calls of `random` / `np.random`, same order) + the kernel must give the same
    float32 tensor bit for bit: crop box, BILINEAR / NEAREST resize, flip, RandomGray, ColorJitter, ToTensor, Normalize, layout."""
    import random
    from dpc_amd.data import draw_k400, draw_ucf101, recipe_to_input
    g = np.load(os.path.join(golden_dir, "aug.npz"))
    F, H0, W0, N, SL, ds, size, crop, start = (int(v) for v in g["params"])
    random.seed(seed)
    np.random.seed(seed)
    jit = None if name.endswith("_geo") else dict(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.25, p=1.0)
    if name.startswith("k400"):
        clip = draw_k400(W0, H0, size, N * SL, jitter=jit)
    else:
        clip = draw_ucf101(W0, H0, crop, size, N * SL, jitter=jit)
    frames = torch.from_numpy(g["frames"]).unsqueeze(0).to(dev)
    block = torch.empty(1, N, 3, SL, size, size, device=dev)
    s2d = torch.empty(N, SL, size // 2, size // 2, 16, device=dev)
    recipe_to_input(lib, frames, [start], [clip], N, SL, ds, size, block, s2d)
    want = torch.from_numpy(g[f"{name}::{seed}"])
    got = block[0].cpu()
    assert torch.equal(got, want), f"{name} seed {seed}: {(got != want).sum().item()} of {want.numel()} values differ, max {(got - want).abs().max().item():.3g}"
    chk = torch.empty_like(s2d)
    lib.call("dpc_pack_input_s2d", block, chk, L.F32, N, SL, size, size, lib.stream())
    assert torch.equal(s2d.cpu(), chk.cpu())
    return clip


@pytest.mark.parametrize("name", ["k400_geo", "ucf101_geo", "k400", "ucf101"])
def test_recipes_match_the_reference_classes_emu(golden_dir, name):
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    seen = set()
    for seed in (1, 2, 3, 4, 5, 6):
        clip = _golden_recipe_case(L.load_emulator(), "cpu", golden_dir, name, seed)
        seen.add((clip["flip"] != 0, bool((clip["gray"] >= 0).any()), tuple(clip["jitter"][0].order)))
    assert len({s[0] for s in seen}) == 2 and len({s[2] for s in seen}) >= (1 if name.endswith("_geo") else 3)   # both flip states, several jitter orders


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["k400_geo", "ucf101_geo", "k400", "ucf101"])
def test_recipes_match_the_reference_classes_gpu(golden_dir, name):
    for seed in (1, 2, 3, 4, 5, 6):
        _golden_recipe_case(L.load_hip(), "cuda:0", golden_dir, name, seed)


def test_resample_tables_match_pil():
    """dpc_amd.data.resample_tables + the kernel's two-pass fixed-point resampling == PIL's Image.resize(BILINEAR) on random images
    (up- and down-scaling, odd sizes), through a crop box"""
    from dpc_amd.data import FrameJitter, recipe_to_input, resample_tables
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    lib = L.load_emulator()
    rng = np.random.default_rng(9)
    for (H0, W0, x1, y1, w, h, size) in ((60, 80, 7, 3, 51, 44, 24), (40, 50, 0, 0, 50, 40, 32), (30, 30, 5, 6, 11, 13, 28)):
        fr = rng.integers(0, 256, (1, 2, H0, W0, 3), dtype=np.uint8)
        xb, xk = resample_tables(w, size)
        yb, yk = resample_tables(h, size)
        jit = (FrameJitter * 2)()
        for j in jit:
            j.order[:] = [255] * 4
        clip = dict(x1=x1, y1=y1, flip=0, xb=xb, xk=xk, yb=yb, yk=yk, gray=np.full(2, -1, np.int8), jitter=jit)
        block = torch.empty(1, 2, 3, 1, size, size)
        recipe_to_input(lib, torch.from_numpy(fr), [0], [clip], 2, 1, 1, size, block, None)
        for n in range(2):
            ref = np.array(Image.fromarray(fr[0, n]).crop((x1, y1, x1 + w, y1 + h)).resize((size, size), Image.BILINEAR))
            t = torch.from_numpy(ref).permute(2, 0, 1).float().div(255)
            t = (t - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
            assert torch.equal(block[0, n, :, 0], t)


def test_bad_draws_are_rejected_on_the_host():
    """the kernel trusts the per-clip draws; frames_to_input / recipe_to_input validate them (a short clip or a box
    outside the frame read out of bounds silently)"""
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    lib = L.load_emulator()
    fr = torch.zeros(1, 10, 20, 24, 3, dtype=torch.uint8)
    block = torch.empty(1, 2, 3, 2, 16, 16)
    ok = torch.tensor([[0, 8, 4, 0]], dtype=torch.int32)
    frames_to_input(lib, fr, ok, None, 2, 2, 3, 16, block, None)
    for bad in ([[1, 8, 4, 0]], [[0, 9, 4, 0]], [[0, 8, 5, 0]], [[-1, 0, 0, 0]]):   # last frame 1 + 9 = 10 >= F; box leaves the frame
        with pytest.raises(ValueError):
            frames_to_input(lib, fr, torch.tensor(bad, dtype=torch.int32), None, 2, 2, 3, 16, block, None)


# ---- the caller: python -m dpc_amd.main --frames (replaces dpc/dataset_3d.py:97-111 + the DataLoader of dpc/main.py:304-313)
def _frames_entry_case(lib, dev_name, tmp_path, golden_dir, dtype, widths, B, size, steps):
    """`main --frames` for one epoch of `steps` batches == the same batches fed by hand as f32 blocks: FrameSource's draws replayed
    with the same seeds -> recipe_to_input(block=...) -- the block tests/golden/aug.npz pins bit for bit to the reference's own
    transform classes -- -> engine.train_step(block).  Parameters and Adam moments after the epoch must be bit-identical."""
    import random
    from dpc_amd import main as dpc_main
    from dpc_amd.data import FrameSource, recipe_to_input
    from dpc_amd.engine import DPCEngine
    from dpc_amd.model import DPC_RNN
    g = np.load(os.path.join(golden_dir, "aug.npz"))
    base = g["frames"]                                            # [14, 60, 80, 3] u8: the frames the reference's transform saw
    rng = np.random.default_rng(3)
    clips = np.stack([np.roll(base, k, axis=0) if k % 2 == 0 else base[:, ::-1][:, :, ::-1].copy() for k in range(B * steps)])
    clips = (clips.astype(np.int16) + rng.integers(-3, 4, clips.shape)).clip(0, 255).astype(np.uint8)
    path = os.path.join(str(tmp_path), "clips.npy")
    np.save(path, clips)
    N, SL, P, ds = 4, 3, 1, 1                                    # 12 of the 14 frames
    pr = os.path.join(str(tmp_path), "probe"); os.makedirs(pr, exist_ok=True)
    argv = ["--net", "resnet18", "--img_dim", str(size), "--batch_size", str(B), "--gpu", "0", "--print_freq", "1", "--dtype", dtype,
            "--num_seq", str(N), "--seq_len", str(SL), "--pred_step", str(P), "--ds", str(ds), "--epochs", "1", "--dataset", "ucf101",
            "--crop", "56", "--frames", path]
    sim = lib if lib.kind != "hip" else None
    dpc_main.main(argv, _simulator=sim, _widths=widths, _probe=pr)
    got = torch.load(os.path.join(pr, "rank0.pt"))
    assert got["step"] == steps
    # by hand: same seeds (dpc_amd.main seeds torch 0, random / np.random with the rank), same draws, but through an f32 block
    torch.manual_seed(0)
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    from dpc_amd.plan import LAYER_WIDTH
    eng = DPCEngine("resnet18", size, N, SL, P, B, dev_name, cdt, widths or LAYER_WIDTH, lib=sim, seed=233)
    init = DPC_RNN(size, N, SL, P, "resnet18", widths=widths or LAYER_WIDTH, seed=0)
    eng.load_params({k: v.detach() for k, v in init.named_parameters()})
    random.seed(0); np.random.seed(0)
    src = FrameSource(path, "ucf101", N, SL, ds, size, B, crop=56)
    assert len(src) == steps and src.span == N * SL
    n = 0
    for frames, starts, cl in src.epoch(torch.device(dev_name)):
        block = torch.empty(B, N, 3, SL, size, size, device=dev_name)
        recipe_to_input(eng.lib, frames, starts, cl, N, SL, src.ds, size, block, None)
        assert torch.isfinite(block).all() and block.std() > 0.3
        eng.train_step(block)
        n += 1
    assert n == steps
    if dev_name != "cpu":
        torch.cuda.synchronize()
    assert torch.equal(got["flat_p"], eng.flat_p.cpu()) and torch.equal(got["flat_m"], eng.flat_m.cpu())


def test_main_frames_entry_emu(tmp_path, golden_dir):
    subprocess.run(["make", "-s", "-j8", "emu"], cwd=ROOT, check=True)
    _frames_entry_case(L.load_emulator(), "cpu", tmp_path, golden_dir, "f32", (8, 16, 32, 32), 1, 64, 2)


@pytest.mark.gpu
def test_main_frames_entry_gpu(tmp_path, golden_dir):
    _frames_entry_case(L.load_hip(), "cuda:0", tmp_path, golden_dir, "bf16", None, 2, 64, 2)
