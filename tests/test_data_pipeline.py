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
