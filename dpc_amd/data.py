"""GPU-side input pipeline: the per-clip random draws of the reference's loader stay on the host
(a handful of integers), the per-pixel work -- frame gather, crop, flip, channel splitting, ToTensor, Normalize, the
[N, C, SL, H, W] layout -- is one HIP kernel (csrc/input_pipeline.hip) that can write the stem's operand directly.

Restates dpc/dataset_3d.py:85-111 (idx_sampler, __getitem__) and the exact-arithmetic transforms of utils/augmentation.py
(RandomCrop :99-143, RandomHorizontalFlip :198-222, RandomGray :224-251, ToTensor / Normalize :368-379).  The resize
of RandomSizedCrop (BILINEAR) and ColorJitter are PIL resampling / colour-space code and are not covered: a run that uses them
keeps them on the host and hands the result over as frames.  Scale with its default NEAREST interpolation IS covered (tables
from PIL itself, `nearest_tables`)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L

MEAN = (0.485, 0.456, 0.406)  # utils/augmentation.py:374
STD = (0.229, 0.224, 0.225)


class ClipAug(C.Structure):
    """struct dpc_clip_aug (include/dpc_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in ("start", "x1", "y1", "flip")]


def draw_clip_params(rng: np.random.Generator, B: int, vlen: int, num_seq: int, seq_len: int, ds: int, H0: int, W0: int, size: int,
                     flip_p: float = 0.5, gray_p: float = 0.5, flip_code: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """the loader's random choices for B clips: start frame (idx_sampler, dataset_3d.py:85-92), crop corner
    (RandomCrop consistent=True), flip (RandomHorizontalFlip consistent=True), per-frame gray channel
    (RandomGray consistent=False, p; -1 = keep colour).  Returns (aug int32 [B,4], gray int8 [B, num_seq*seq_len])."""
    span = vlen - num_seq * seq_len * ds
    if span <= 0:
        raise ValueError("video too short (dataset_3d.py:87 drops it)")
    aug = np.zeros((B, 4), np.int32)
    aug[:, 0] = rng.integers(0, span, B)
    aug[:, 1] = rng.integers(0, W0 - size + 1, B)   # `size` here = the side of the crop box (224 in the ucf101 recipe)
    aug[:, 2] = rng.integers(0, H0 - size + 1, B)
    aug[:, 3] = (rng.random(B) < flip_p) * flip_code
    gray = np.where(rng.random((B, num_seq * seq_len)) < gray_p, rng.integers(0, 3, (B, num_seq * seq_len)), -1).astype(np.int8)
    return aug, gray


_NEAREST_TABLES = {}


def nearest_tables(crop: int, size: int):
    """Scale(size=(size, size)) with its default NEAREST interpolation (utils/augmentation.py:20-43; the ucf101 recipe crops
    224 and scales to img_dim, dpc/main.py:116-118): output index -> index inside the crop, obtained from PIL itself by resizing
    a coordinate ramp, so whatever rounding PIL's resampler applies is reproduced exactly."""
    key = (crop, size)
    if key not in _NEAREST_TABLES:
        from PIL import Image
        ramp = Image.fromarray(np.tile(np.arange(crop, dtype=np.int32), (2, 1)), mode="I")
        xt = np.array(ramp.resize((size, 2), Image.NEAREST))[0].astype(np.int32)
        _NEAREST_TABLES[key] = xt
    return _NEAREST_TABLES[key]


def frames_to_input(lib: L.Lib, frames: torch.Tensor, aug: torch.Tensor, gray: Optional[torch.Tensor], num_seq: int, seq_len: int,
                    ds: int, size: int, block: Optional[torch.Tensor] = None, s2d: Optional[torch.Tensor] = None, crop: Optional[int] = None):
    """frames u8 [B,F,H0,W0,3]; aug int32 [B,4] (start, x1, y1, flip: 0 / 1 after crop (k400) / 2 before crop (ucf101));
    gray int8 [B, N*SL] or None -- all on the kernels' device.  crop: side of the crop box when it differs from `size`
    (Scale, NEAREST).  Fills block f32 [B,N,3,SL,size,size] and / or s2d [B*N,SL,size/2,size/2,16] (compute dtype)."""
    B, F, H0, W0, ch = frames.shape
    if ch != 3 or frames.dtype != torch.uint8 or aug.dtype != torch.int32 or tuple(aug.shape) != (B, 4):
        raise ValueError("frames must be uint8 [B,F,H0,W0,3], aug int32 [B,4]")
    mean, std = (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD)
    tab = None
    if crop is not None and crop != size:
        tab = torch.from_numpy(nearest_tables(crop, size)).to(frames.device)
    lib.call("dpc_frames_to_input", frames.contiguous(), B, F, H0, W0, aug.contiguous(), gray.contiguous() if gray is not None else None,
             num_seq, seq_len, ds, size, size, tab, tab, crop or size, crop or size, mean, std, block, s2d,
             L.dtype_code(s2d.dtype) if s2d is not None else L.F32, lib.stream())
    return block, s2d
