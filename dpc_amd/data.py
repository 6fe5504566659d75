"""GPU-side input pipeline: the per-clip random draws of the reference's loader stay on the host
(a handful of integers), the per-pixel work -- frame gather, crop, flip, channel splitting, ToTensor, Normalize, the
[N, C, SL, H, W] layout -- is one HIP kernel (csrc/input_pipeline.hip) that can write the stem's operand directly.

Restates dpc/dataset_3d.py:85-111 (idx_sampler, __getitem__) and the transforms of utils/augmentation.py: RandomCrop :99-143,
RandomSizedCrop :144-196 (crop box + PIL's BILINEAR resize as fixed-point resampling tables, `resample_tables`),
RandomHorizontalFlip :198-222, RandomGray :224-251, ColorJitter :253-351 (per-frame factors and shuffled order; the PIL /
torchvision arithmetic lives in the kernel), Scale :20-43 (NEAREST tables from PIL itself, `nearest_tables`), ToTensor /
Normalize :368-379.  `draw_k400` / `draw_ucf101` make the random choices of the two training recipes of dpc/main.py:114-132
with the SAME calls of `random` / `np.random` in the same order as the reference's classes, so a seeded run reproduces the
reference's clip bit for bit (tests/golden/aug.npz holds clips produced by the reference's own classes)."""
from __future__ import annotations

import ctypes as C
import math
import random
from typing import List, Optional, Tuple

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
    # the kernel trusts the draws: check them here (a short clip or a bad box would read outside the frames buffer)
    a = aug.detach().cpu()
    box = crop or size
    if int(a[:, 0].min()) < 0 or int(a[:, 0].max()) + (num_seq * seq_len - 1) * ds >= F:
        raise ValueError(f"sampled frames leave the video: start up to {int(a[:, 0].max())}, {num_seq * seq_len} frames every {ds}, {F} available")
    if int(a[:, 1].min()) < 0 or int(a[:, 2].min()) < 0 or int(a[:, 1].max()) + box > W0 or int(a[:, 2].max()) + box > H0:
        raise ValueError(f"crop box of {box} pixels leaves the {W0} x {H0} frame")
    mean, std = (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD)
    tab = None
    if crop is not None and crop != size:
        tab = torch.from_numpy(nearest_tables(crop, size)).to(frames.device)
    lib.call("dpc_frames_to_input", frames.contiguous(), B, F, H0, W0, aug.contiguous(), gray.contiguous() if gray is not None else None,
             num_seq, seq_len, ds, size, size, tab, tab, crop or size, crop or size, mean, std, block, s2d,
             L.dtype_code(s2d.dtype) if s2d is not None else L.F32, lib.stream())
    return block, s2d


# ------------------------------------------------------------------------------------------------------------------
# the full training recipes of dpc/main.py:114-132 (ColorJitter and the BILINEAR resize of RandomSizedCrop included)
# ------------------------------------------------------------------------------------------------------------------
RS_PREC = 32 - 8 - 2  # PIL ImagingResample PRECISION_BITS (8 bits per channel)


class FrameJitter(C.Structure):
    """struct dpc_frame_jitter (include/dpc_hip.h)"""
    _fields_ = [("factor", C.c_float * 3), ("hue_shift", C.c_int32), ("order", C.c_uint8 * 4)]


def resample_tables(in_size: int, out_size: int, first: int = 0, count: Optional[int] = None, support: float = 1.0):
    """PIL's ImagingResample coefficient tables for resizing a line of `in_size` samples to `out_size` with the BILINEAR (triangle)
    filter, restated from libImaging/Resample.c (precompute_coeffs + normalize_coeffs_8bpc): double-precision weights,
    normalised, rounded to 22-bit fixed point.  Rows [first, first + count) of the table.  Returns (bounds int32 [count, 2] =
    (first source index, taps), coefficients int32 [count, ksize])."""
    count = out_size - first if count is None else count
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    sup = support * fscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((count, 2), np.int32)
    coef = np.zeros((count, ksize), np.int32)
    ss = 1.0 / fscale
    for i in range(count):
        xx = first + i
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = sum(w)  # same left-to-right accumulation as the C loop
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[i] = (xmin, xmax)
        for x, v in enumerate(w):
            coef[i, x] = int(-0.5 + v * (1 << RS_PREC)) if v < 0 else int(0.5 + v * (1 << RS_PREC))
    return bounds, coef


def _tables_from_index(idx: np.ndarray):
    """a pure index map (NEAREST) in resampling-table form: one tap of weight 1.0 (clip8((1 << 21) + v << 22) == v)"""
    b = np.stack([idx.astype(np.int32), np.ones_like(idx, dtype=np.int32)], 1)
    return b, np.full((len(idx), 1), 1 << RS_PREC, np.int32)


def _draw_gray_and_jitter(n_frames: int, gray_p: float, jitter: Optional[dict]):
    """RandomGray(consistent=False, p) then ColorJitter(..., consistent=False, p) over the frames of one clip, call for call
    (utils/augmentation.py:229-243 and :310-351)"""
    gray = np.full(n_frames, -1, np.int8)
    for i in range(n_frames):                       # RandomGray: one Bernoulli per frame, then np.random.choice(3)
        if random.random() < gray_p:
            gray[i] = np.random.choice(3)
    jit = (FrameJitter * n_frames)()
    for i in range(n_frames):
        jit[i].order[:] = [255, 255, 255, 255]
    if jitter is not None and random.random() < jitter.get("p", 1.0):
        for i in range(n_frames):                   # get_params per frame: four uniform draws, then random.shuffle of the list
            ops: List[int] = []
            for op, name in enumerate(("brightness", "contrast", "saturation", "hue")):
                v = jitter.get(name, 0)
                if not v:
                    continue
                lo, hi = (-v, v) if name == "hue" else (max(0.0, 1.0 - v), 1.0 + v)   # _check_input
                f = random.uniform(lo, hi)
                if name == "hue":
                    jit[i].hue_shift = int(f * 255) & 0xFF   # np.uint8(hue_factor * 255): C truncation, uint8 wrap
                else:
                    jit[i].factor[op] = f
                ops.append(op)
            random.shuffle(ops)
            for k, op in enumerate(ops):
                jit[i].order[k] = op
    return gray, jit


def draw_k400(W0: int, H0: int, size: int, n_frames: int, gray_p: float = 0.5,
              jitter: Optional[dict] = dict(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.25, p=1.0)):
    """the k400 training recipe (dpc/main.py:124-132) for ONE clip of n_frames frames of W0 x H0 pixels:
    RandomSizedCrop(size, consistent=True, p=1.0) -> RandomHorizontalFlip(consistent=True) -> RandomGray(p=0.5) -> ColorJitter.
    Consumes `random` / `np.random` exactly as the reference's classes do.  Returns dict(x1, y1, flip, xb, xk, yb, yk, gray, jitter)."""
    x1 = y1 = 0
    xt = yt = None
    if random.random() < 1.0:                                   # RandomSizedCrop.threshold = p = 1.0
        for _ in range(10):
            area = W0 * H0
            target_area = random.uniform(0.5, 1) * area
            aspect = random.uniform(3. / 4, 4. / 3)
            w = int(round(math.sqrt(target_area * aspect)))
            h = int(round(math.sqrt(target_area / aspect)))
            if random.random() < 0.5:
                w, h = h, w
            if w <= W0 and h <= H0:
                x1 = random.randint(0, W0 - w)
                y1 = random.randint(0, H0 - h)
                xt, yt = resample_tables(w, size), resample_tables(h, size)
                break
        else:   # fallback (augmentation.py:190-193): Scale(size) then CenterCrop(size), both on the whole frame
            if W0 < H0:
                ow, oh = size, int(size * H0 / W0)
            else:
                oh, ow = size, int(size * W0 / H0)
            cx, cy = int(round((ow - size) / 2.)), int(round((oh - size) / 2.))
            xt, yt = resample_tables(W0, ow, cx, size), resample_tables(H0, oh, cy, size)
    flip = 1 if random.random() < 0.5 else 0                    # RandomHorizontalFlip(consistent=True): after the crop + resize
    gray, jit = _draw_gray_and_jitter(n_frames, gray_p, jitter)
    return dict(x1=x1, y1=y1, flip=flip, xb=xt[0], xk=xt[1], yb=yt[0], yk=yt[1], gray=gray, jitter=jit)


def draw_ucf101(W0: int, H0: int, crop: int, size: int, n_frames: int, gray_p: float = 0.5,
                jitter: Optional[dict] = dict(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.25, p=1.0)):
    """the ucf101 training recipe (dpc/main.py:114-123): RandomHorizontalFlip -> RandomCrop(crop) -> Scale((size, size)) [NEAREST]
    -> RandomGray(p=0.5) -> ColorJitter, same contract as draw_k400 (flip code 2: the frame is flipped BEFORE the crop)"""
    flip = 2 if random.random() < 0.5 else 0
    x1 = y1 = 0
    if not (W0 == crop and H0 == crop):                         # RandomCrop draws only when there is something to crop
        x1 = random.randint(0, W0 - crop)
        y1 = random.randint(0, H0 - crop)
    idx = nearest_tables(crop, size) if crop != size else np.arange(size, dtype=np.int32)
    xb, xk = _tables_from_index(idx)
    gray, jit = _draw_gray_and_jitter(n_frames, gray_p, jitter)
    return dict(x1=x1, y1=y1, flip=flip, xb=xb, xk=xk, yb=xb.copy(), yk=xk.copy(), gray=gray, jitter=jit)


def recipe_to_input(lib: L.Lib, frames: torch.Tensor, starts, clips: "List[dict]", num_seq: int, seq_len: int, ds: int, size: int,
                    block: Optional[torch.Tensor] = None, s2d: Optional[torch.Tensor] = None):
    """frames u8 [B,F,H0,W0,3] on the kernels' device + one draw_k400 / draw_ucf101 result per clip -> block f32 [B,N,3,SL,size,size]
    and / or the stem's space-to-depth operand, through dpc_frames_to_input_ex.  starts[b] = first sampled frame of clip b."""
    B, F, H0, W0, ch = frames.shape
    n_fr = num_seq * seq_len
    if ch != 3 or frames.dtype != torch.uint8 or len(clips) != B:
        raise ValueError("frames must be uint8 [B,F,H0,W0,3] with one recipe draw per clip")
    dev = frames.device
    ksx = max(c["xk"].shape[1] for c in clips)
    ksy = max(c["yk"].shape[1] for c in clips)
    aug = np.zeros((B, 4), np.int32)
    xb, yb = np.zeros((B, size, 2), np.int32), np.zeros((B, size, 2), np.int32)
    xk, yk = np.zeros((B, size, ksx), np.int32), np.zeros((B, size, ksy), np.int32)
    gray = np.zeros((B, n_fr), np.int8)
    jit = (FrameJitter * (B * n_fr))()
    any_jitter = False
    for b, c in enumerate(clips):
        # host-side validation: the kernel trusts these: sampled frames inside the video, source windows inside the frame
        if starts[b] < 0 or starts[b] + (n_fr - 1) * ds >= F:
            raise ValueError(f"clip {b}: frames {starts[b]} .. {starts[b] + (n_fr - 1) * ds} do not fit a video of {F} frames")
        if c["x1"] < 0 or c["y1"] < 0 or c["x1"] + int((c["xb"][:, 0] + c["xb"][:, 1]).max()) > W0 or \
                c["y1"] + int((c["yb"][:, 0] + c["yb"][:, 1]).max()) > H0:
            raise ValueError(f"clip {b}: the crop box / resampling window leaves the {W0} x {H0} frame")
        aug[b] = (starts[b], c["x1"], c["y1"], c["flip"])
        xb[b], yb[b] = c["xb"], c["yb"]
        xk[b, :, :c["xk"].shape[1]] = c["xk"]
        yk[b, :, :c["yk"].shape[1]] = c["yk"]
        gray[b] = c["gray"]
        for i in range(n_fr):
            jit[b * n_fr + i] = c["jitter"][i]
            any_jitter |= c["jitter"][i].order[0] != 255
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(aug=aug, xb=xb, xk=xk, yb=yb, yk=yk, gray=gray).items()}
    rs = L.Resample(t["xb"].data_ptr(), t["xk"].data_ptr(), t["yb"].data_ptr(), t["yk"].data_ptr(), ksx, ksy)
    jt = u8 = ls = None
    if any_jitter:
        jt = torch.frombuffer(bytearray(bytes(jit)), dtype=torch.uint8).clone().to(dev)
        u8 = torch.empty(B * n_fr * size * size * 3, dtype=torch.uint8, device=dev)
        ls = torch.empty(B * n_fr, dtype=torch.int64, device=dev)
    mean, std = (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD)
    lib.call("dpc_frames_to_input_ex", frames.contiguous(), B, F, H0, W0, t["aug"], t["gray"], num_seq, seq_len, ds, size, size, C.byref(rs),
             jt, u8, ls, mean, std, block, s2d, L.dtype_code(s2d.dtype) if s2d is not None else L.F32, lib.stream())
    return block, s2d


class FrameSource:
    """Decoded clips as ONE uint8 array [clips, F, H0, W0, 3] (``np.load(path, mmap_mode='r')`` or an array) -- the data source of
    ``python -m dpc_amd.main --frames``.  It stands where the reference has ``dataset[index]`` under
    ``DataLoader(RandomSampler(dataset), batch_size, drop_last=True)`` (dpc/dataset_3d.py:88-111, dpc/main.py:304-313): per
    epoch a random permutation of the clips cut into batches (the last partial one dropped); per clip the start frame of
    ``idx_sampler`` (``np.random.choice(range(vlen - num_seq * seq_len * ds), 1)``, dataset_3d.py:88-92) and then the draws of the
    training transform (``draw_k400`` / ``draw_ucf101``: the ``Compose`` of dpc/main.py:114-132, same ``random`` / ``np.random``
    calls in the same order).  What it hands out is NOT a float tensor: the uint8 frames the clip needs (its span of
    (num_seq * seq_len - 1) * ds + 1 frames) and the draws; ``DPCEngine.load_recipe`` runs crop / resize / flip / grey / jitter /
    ToTensor / Normalize / layout on the GPU and fills the stem's operand.  Readers of video files are out of scope (SURVEY section 2
    row 8); anything that can produce this array (a decoder, a frame cache) plugs in here."""

    def __init__(self, frames, dataset: str, num_seq: int, seq_len: int, ds: int, size: int, batch: int, rank: int = 0, world: int = 1,
                 crop: int = 224):
        self.frames = np.load(frames, mmap_mode="r") if isinstance(frames, (str, bytes)) or hasattr(frames, "__fspath__") else frames
        fr = self.frames
        if fr.ndim != 5 or fr.shape[4] != 3 or fr.dtype != np.uint8:
            raise ValueError("--frames wants a uint8 array [clips, F, H0, W0, 3]")
        if dataset not in ("k400", "ucf101"):
            raise ValueError("dataset not supported")   # dpc/main.py:301
        self.dataset, self.N, self.SL, self.size, self.batch, self.rank, self.world, self.crop = dataset, num_seq, seq_len, size, batch, rank, world, crop
        self.ds = 5 if dataset == "k400" else ds           # dpc/main.py:294 (k400: downsample=5), :299 (ucf101: args.ds)
        self.n_fr = num_seq * seq_len
        self.span = (self.n_fr - 1) * self.ds + 1
        if fr.shape[1] - self.n_fr * self.ds <= 0:
            raise ValueError(f"clips of {fr.shape[1]} frames are too short for {num_seq} x {seq_len} frames at stride {self.ds} "
                             "(dpc/dataset_3d.py:80-82 drops such videos)")
        if dataset == "ucf101" and (fr.shape[2] < crop or fr.shape[3] < crop):
            raise ValueError(f"the ucf101 recipe crops {crop} x {crop} (dpc/main.py:117): frames of {fr.shape[3]} x {fr.shape[2]} are too small")

    def __len__(self):   # batches per epoch, drop_last
        return self.frames.shape[0] // (self.batch * self.world)

    def epoch(self, device):
        """yields (frames u8 [B, span, H0, W0, 3] on `device`, starts (all 0: the span is cut out on the host), clips) per batch of THIS rank"""
        fr = self.frames
        F, H0, W0 = fr.shape[1:4]
        order = torch.randperm(fr.shape[0]).tolist()     # RandomSampler: torch's generator, seeded by torch.manual_seed(0) (dpc/main.py:50)
        gb = self.batch * self.world
        for i in range(len(self)):
            mine = order[i * gb + self.rank * self.batch: i * gb + (self.rank + 1) * self.batch]
            clips, host = [], np.empty((self.batch, self.span, H0, W0, 3), np.uint8)
            for j, ci in enumerate(mine):
                start = int(np.random.choice(range(F - self.n_fr * self.ds), 1)[0])         # idx_sampler
                host[j] = fr[ci, start:start + self.span]
                clips.append(draw_k400(W0, H0, self.size, self.n_fr) if self.dataset == "k400"
                             else draw_ucf101(W0, H0, self.crop, self.size, self.n_fr))
            yield torch.from_numpy(host).to(device, non_blocking=False), [0] * self.batch, clips
