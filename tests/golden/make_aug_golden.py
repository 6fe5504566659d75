"""Generate tests/golden/aug.npz by running THE REFERENCE'S OWN augmentation classes (utils/augmentation.py) on seeded inputs.

Runs only in the build container (needs /root/reference and PIL).  What the GPU-side input pipeline (csrc/input_pipeline.hip,
dpc_amd/data.py) is held to: the two training recipes of dpc/main.py:114-132, composed exactly as there, applied to a small
synthetic video under `random.seed(s); np.random.seed(s)`; the fixture stores the video, the recipe parameters, the seeds and the
clips the reference produced ([N, 3, SL, H, W] f32 after dataset_3d.py:107-111's stack / view / transpose).

Shims applied here and nowhere else:
  * `torchvision` is not installed (no network).  utils/augmentation.py uses it for transforms.Compose / Lambda / ToTensor /
    Normalize and for transforms.functional.adjust_{brightness,contrast,saturation,hue}.  The stub below restates those few
    functions from torchvision's published source (torchvision/transforms/_functional_pil.py, transforms.py, v0.2 .. v0.19 agree):
    the adjust_* functions are PIL ImageEnhance.{Brightness,Contrast,Color} and an HSV round trip with `np.uint8(hue_factor * 255)`
    added to H (uint8 wrap-around); ToTensor = HWC uint8 -> CHW float / 255; Normalize = (t - mean) / std in f32.  PIL itself IS
    installed, so every pixel operation below the stub is the real library.
  * collections.Iterable (removed in Python 3.10) -> collections.abc.Iterable, used by Scale.__init__ (augmentation.py:22).

Also re-verified here, exhaustively, before anything is written: the kernel's restatement of PIL's RGB -> L, RGB <-> HSV for all
2^24 colours and of Image.blend for all 65 536 value pairs (numpy twins of the device code in csrc/input_pipeline.hip).

usage:  python tests/golden/make_aug_golden.py
"""
import collections
import collections.abc
import os
import random
import sys
import types

import numpy as np
import torch
from PIL import Image, ImageEnhance

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
collections.Iterable = collections.abc.Iterable


# ---- torchvision stub (restated from its published source; PIL does the work) -------------------------------------------------
def _adjust_hue(img, hue_factor):
    if not (-0.5 <= hue_factor <= 0.5):
        raise ValueError("hue_factor is not in [-0.5, 0.5].")
    mode = img.mode
    if mode in {"L", "1", "I", "F"}:
        return img
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = (np_h.astype(np.int32) + (int(hue_factor * 255) & 0xFF)).astype(np.uint8)   # np_h += np.uint8(hue_factor * 255), uint8 wrap
    h = Image.fromarray(np_h, "L")
    return Image.merge("HSV", (h, s, v)).convert(mode)


class _Lambda:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)


class _Compose:
    def __init__(self, ts):
        self.transforms = ts

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _ToTensor:
    def __call__(self, pic):
        a = torch.from_numpy(np.array(pic, np.uint8, copy=True))
        return a.view(pic.size[1], pic.size[0], 3).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        m = torch.as_tensor(self.mean, dtype=t.dtype)
        s = torch.as_tensor(self.std, dtype=t.dtype)
        return t.clone().sub_(m.view(-1, 1, 1)).div_(s.view(-1, 1, 1))


tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tvf = types.ModuleType("torchvision.transforms.functional")
tvf.adjust_brightness = lambda img, f: ImageEnhance.Brightness(img).enhance(f)
tvf.adjust_contrast = lambda img, f: ImageEnhance.Contrast(img).enhance(f)
tvf.adjust_saturation = lambda img, f: ImageEnhance.Color(img).enhance(f)
tvf.adjust_hue = _adjust_hue
tvt.Lambda, tvt.Compose, tvt.ToTensor, tvt.Normalize, tvt.functional = _Lambda, _Compose, _ToTensor, _Normalize, tvf
tv.transforms = tvt
sys.modules["torchvision"], sys.modules["torchvision.transforms"], sys.modules["torchvision.transforms.functional"] = tv, tvt, tvf

sys.path.insert(0, f"{REF}/utils")
import augmentation as A  # noqa: E402  (the reference's own classes)


# ---- exhaustive checks of the kernel's PIL restatements (numpy twins of csrc/input_pipeline.hip) ------------------------------
def verify_pil_arithmetic():
    a = np.arange(1 << 24, dtype=np.uint32)
    cols = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    r, g, b = (cols.reshape(-1, 3)[:, i].astype(np.int64) for i in range(3))
    f32 = np.float32
    # L
    L = np.array(Image.fromarray(cols, "RGB").convert("L")).reshape(-1)
    assert np.array_equal(L, (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16)
    # RGB -> HSV
    hsv = np.array(Image.fromarray(cols, "RGB").convert("HSV")).reshape(-1, 3)
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    with np.errstate(all="ignore"):
        cr = (maxc - minc).astype(f32)
        s = cr / maxc.astype(f32)
        rc, gc, bc = ((maxc - c).astype(f32) / cr for c in (r, g, b))
        d = np.float64
        h = np.where(r == maxc, (bc - gc).astype(d), np.where(g == maxc, 2.0 + rc.astype(d) - bc.astype(d), 4.0 + gc.astype(d) - rc.astype(d))).astype(f32)
        h = np.fmod(h.astype(d) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.clip((h.astype(d) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(d) * 255.0).astype(np.int64), 0, 255)
    gray = minc == maxc
    assert np.array_equal(np.where(gray, 0, uh), hsv[:, 0]) and np.array_equal(np.where(gray, 0, us), hsv[:, 1]) and np.array_equal(maxc, hsv[:, 2])
    # HSV -> RGB
    rgb = np.array(Image.fromarray(cols, "HSV").convert("RGB")).reshape(-1, 3)
    hh, ss, vv = r, g, b
    hf = hh.astype(f32).astype(d) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int64)
    f = (hf - i.astype(f32).astype(d)).astype(f32).astype(d)
    fs = (ss.astype(f32).astype(d) / 255.0).astype(f32).astype(d)
    cround = lambda x: np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)  # noqa: E731
    vd = vv.astype(d)
    p, q, t = (np.clip(cround(vd * e), 0, 255) for e in (1.0 - fs, 1.0 - fs * f, 1.0 - fs * (1.0 - f)))
    sel = i % 6
    R, G, B = np.choose(sel, [vv, q, p, p, t, vv]), np.choose(sel, [t, vv, vv, q, p, p]), np.choose(sel, [p, p, t, vv, vv, q])
    for got, ref in ((R, rgb[:, 0]), (G, rgb[:, 1]), (B, rgb[:, 2])):
        assert np.array_equal(np.where(ss == 0, vv, got), ref)
    # blend
    v = np.arange(256, dtype=np.uint8)
    g1, g2 = np.meshgrid(v, v, indexing="ij")
    i1, i2 = Image.fromarray(g1, "L"), Image.fromarray(g2, "L")
    rng = np.random.default_rng(0)
    for fac in list(rng.uniform(0, 2, 60)) + [0.0, 1.0, 0.5, 1.5, 2.0]:
        al = f32(fac)
        tt = g1.astype(f32) + al * (g2.astype(np.int32) - g1.astype(np.int32)).astype(f32)
        mine = np.trunc(tt) if 0 <= al <= 1 else np.where(tt <= 0, 0, np.where(tt >= 255, 255, np.trunc(tt)))
        assert np.array_equal(np.array(Image.blend(i1, i2, fac)), mine.astype(np.uint8)), fac
    print("PIL arithmetic restatements: exact for all 2^24 colours (L, RGB<->HSV) and all value pairs (blend)")


def synthetic_video(F, H0, W0, seed):
    """smooth colour gradients + moving blobs + noise: every hue sector, saturated and grey pixels"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H0, 0:W0].astype(np.float32)
    out = np.zeros((F, H0, W0, 3), np.uint8)
    for f in range(F):
        img = np.stack([127 + 120 * np.sin(xx / 9.0 + f * 0.3), 127 + 120 * np.cos(yy / 7.0 - f * 0.2), 127 + 120 * np.sin((xx + yy) / 11.0 + f)], -1)
        img += rng.normal(0, 12, img.shape)
        img[(xx - 20 - 2 * f) ** 2 + (yy - 25) ** 2 < 90] = (250, 10, 5)
        img[(xx - 55) ** 2 + (yy - 15 - f) ** 2 < 60] = (128, 128, 128)
        out[f] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def run_reference(transform, frames, idx, N, SL):
    seq = [Image.fromarray(frames[i]) for i in idx]                     # pil_loader output (dataset_3d.py:104)
    t_seq = transform(seq)
    (C, H, W) = t_seq[0].size()
    t_seq = torch.stack(t_seq, 0)
    return t_seq.view(N, SL, C, H, W).transpose(1, 2).contiguous()     # dataset_3d.py:107-111


def main():
    verify_pil_arithmetic()
    F, H0, W0, N, SL, ds, size, crop = 14, 60, 80, 2, 2, 3, 24, 44
    frames = synthetic_video(F, H0, W0, 0)
    start = 1
    idx = (np.arange(N)[:, None] * ds * SL + start + np.arange(SL)[None, :] * ds).reshape(-1)   # idx_sampler layout (dataset_3d.py:88-92)
    out = {"frames": frames, "params": np.array([F, H0, W0, N, SL, ds, size, crop, start], np.int32)}
    recipes = {
        # dpc/main.py:124-132 with img_dim = size
        "k400": lambda: tvt.Compose([A.RandomSizedCrop(size=size, consistent=True, p=1.0), A.RandomHorizontalFlip(consistent=True),
                                     A.RandomGray(consistent=False, p=0.5),
                                     A.ColorJitter(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.25, p=1.0), A.ToTensor(), A.Normalize()]),
        # dpc/main.py:115-123 with the 224 crop scaled down to `crop`
        "ucf101": lambda: tvt.Compose([A.RandomHorizontalFlip(consistent=True), A.RandomCrop(size=crop, consistent=True),
                                       A.Scale(size=(size, size)), A.RandomGray(consistent=False, p=0.5),
                                       A.ColorJitter(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.25, p=1.0), A.ToTensor(), A.Normalize()]),
        # the geometric part alone (no ColorJitter): crop / flip / gray / scale pinned without the colour code
        "k400_geo": lambda: tvt.Compose([A.RandomSizedCrop(size=size, consistent=True, p=1.0), A.RandomHorizontalFlip(consistent=True),
                                         A.RandomGray(consistent=False, p=0.5), A.ToTensor(), A.Normalize()]),
        "ucf101_geo": lambda: tvt.Compose([A.RandomHorizontalFlip(consistent=True), A.RandomCrop(size=crop, consistent=True),
                                           A.Scale(size=(size, size)), A.RandomGray(consistent=False, p=0.5), A.ToTensor(), A.Normalize()]),
    }
    seeds = [1, 2, 3, 4, 5, 6]
    for name, make in recipes.items():
        for s in seeds:
            random.seed(s)
            np.random.seed(s)
            out[f"{name}::{s}"] = run_reference(make(), frames, idx, N, SL).numpy()
    out["seeds"] = np.array(seeds, np.int32)
    path = os.path.join(HERE, "aug.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
