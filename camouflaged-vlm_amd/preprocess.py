"""GPU preprocessing in front of the hot path: uint8 HWC image on the device ->
``inp`` (1,3,S,S), ``clip_image`` (1,3,R,R), ``clip_mask`` (1,1,R,R).  ``inp`` and ``clip_image`` are exactly what
demo.py:93-101 / datasets/wrappers.py:22-62 build with torchvision + Pillow on the CPU.  ``clip_mask`` follows the
DATASET WRAPPER by default (a uint8 all-255 mask through ToTensor + Normalize(0.5, 0.26): 1.923, wrappers.py:62, the value
the model is trained and evaluated with); demo.py:103 builds its mask from a float64 array (PIL mode 'F'), which
ToTensor does not divide by 255, so the demo script feeds (255 - 0.5) / 0.26 = 978.8 instead -- `convention="demo"`
reproduces that.  The coefficient tables of
Pillow's resample are computed on the host once per (in, out, filter) and cached; all per-pixel work is in
camouflaged-vlm_amd/csrc/preprocess.hip."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from . import hip

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
OPENAI_MEAN, OPENAI_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
_PRECISION_BITS = 32 - 8 - 2


def nearest_indices(in_size: int, out_size: int) -> np.ndarray:
    """source index of every output pixel of Pillow's NEAREST resize along one axis (see `_coeffs`)"""
    scale = np.float64(in_size) / np.float64(out_size)
    steps = np.full(out_size, scale, dtype=np.float64)
    steps[0] = scale * 0.5
    return np.cumsum(steps).astype(np.int64)


def _coeffs(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow precompute_coeffs + normalize_coeffs_8bpc (libImaging/Resample.c) for every output index at once: the same
    double-precision operations in the same order per row (the window sum runs left to right over the <= ksize taps, one
    vectorised step per tap), so the tables equal the per-index loop bit for bit (tests/test_preprocess.py) -- a new image
    size costs the host ~0.1 ms instead of ~6 ms per table."""
    if filt == "nearest":
        # Pillow's NEAREST resize is ImagingScaleAffine (libImaging/Geometry.c): the source coordinate starts at scale / 2 and is
        # ADVANCED BY REPEATED ADDITION of scale in double (`xo += a[0]`), truncated per output pixel -- not (x + 0.5) * scale, which
        # lands on the other side of an integer for scales like 3.2.  np.cumsum adds left to right: the same doubles (pinned against
        # Pillow on 159 size pairs, tests/test_preprocess.py).  As a resample table: one tap of weight 1.0 (2^22: the value comes back
        # exactly).
        nearest_index = nearest_indices(in_size, out_size)
        idx = np.minimum(nearest_index, in_size - 1)
        return (np.stack([idx, np.ones_like(idx)], axis=1).astype(np.int32),
                np.full((out_size, 1), 1 << _PRECISION_BITS, dtype=np.int32))
    support0 = 1.0 if filt == "bilinear" else 2.0
    scale = filterscale = float(in_size) / out_size
    filterscale = max(filterscale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)               # C's (int) cast: truncation (operands >= 0 here
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin  # or clamped to 0 by the max)
    xmin_neg = (center - support + 0.5) < 0                                         # (int) of a negative value truncates towards 0:
    xmin = np.where(xmin_neg, 0, xmin)                                              # clamped to 0 either way
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    valid = taps < xmax[:, None]
    x = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * ss)
    if filt == "bilinear":
        w = np.where(x < 1.0, 1.0 - x, 0.0)
    else:
        a = -0.5
        w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                     np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
    w = np.where(valid, w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for t in range(ksize):                               # left-to-right double accumulation, as the C loop
        ww = np.where(valid[:, t], ww + w[:, t], ww)
    w = np.where((ww != 0.0)[:, None], w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    scaled = w * float(1 << _PRECISION_BITS)
    kk = np.where(w < 0, (-0.5 + scaled).astype(np.int64), (0.5 + scaled).astype(np.int64))   # (int) truncates towards zero
    kk = np.where(valid, kk, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk


_TABLES: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}      # (n_in, n_out, filter, device) -> device tables, shared by every instance
_TABLE_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def clip_mask_value(convention: str = "wrapper") -> float:
    """pass-1 alpha of an all-ones mask after alpha_clip.py:88-94 (Normalize(0.5, 0.26)): "wrapper" = (1 - 0.5) / 0.26 = 1.923
    (uint8 mask, ToTensor divides by 255, wrappers.py:62); "demo" = (255 - 0.5) / 0.26 = 978.8 (demo.py:102-104: float64 array,
    PIL mode 'F', which ToTensor does not rescale)."""
    if convention not in ("wrapper", "demo"):
        raise ValueError(f"unknown clip_mask convention {convention!r}")
    return (1.0 - 0.5) / 0.26 if convention == "wrapper" else (255.0 - 0.5) / 0.26


class GpuPreprocess:
    def __init__(self, inp_size: int = 1024, clip_size: int = 336, device="cuda"):
        self.S, self.R, self.device = inp_size, clip_size, torch.device(device)
        f = lambda v: torch.tensor(v, dtype=torch.float32, device=self.device)
        self.im_mean, self.im_std, self.cl_mean, self.cl_std = f(IMAGENET_MEAN), f(IMAGENET_STD), f(OPENAI_MEAN), f(OPENAI_STD)

    def _table(self, n_in: int, n_out: int, filt: str):
        key = (n_in, n_out, filt, str(self.device))
        if key not in _TABLES:
            b, k = _coeffs(n_in, n_out, filt)
            # The tables are shared by every stream that preprocesses (the evaluation loop's tail stream resizes ground truths):
            # they are uploaded on a stream of their own, from pinned memory, and the host waits for THAT stream only -- a pageable
            # copy, or a wait on the caller's stream, would cost the host its lead over everything queued there.
            st = _TABLE_STREAMS.get(key[3])
            if st is None:
                st = _TABLE_STREAMS[key[3]] = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(st):
                _TABLES[key] = (torch.from_numpy(b).pin_memory().to(self.device, non_blocking=True),
                                torch.from_numpy(k).pin_memory().to(self.device, non_blocking=True))
            st.synchronize()
        return _TABLES[key]

    def resize(self, img: torch.Tensor, out_h: int, out_w: int, filt: str) -> torch.Tensor:
        """uint8 [N][H][W][C] on the device -> uint8 [N][out_h][out_w][C] (== PIL.Image.resize)."""
        N, H, W, C = img.shape
        x = img.contiguous()
        if out_w != W:
            b, k = self._table(W, out_w, filt)
            y = torch.empty(N, H, out_w, C, dtype=torch.uint8, device=self.device)
            hip.resample_u8(x, b, k, out_w, 1, y)
            x = y
        if out_h != H:
            b, k = self._table(H, out_h, filt)
            y = torch.empty(N, out_h, x.shape[2], C, dtype=torch.uint8, device=self.device)
            hip.resample_u8(x, b, k, out_h, 0, y)
            x = y
        return x

    def sam_input(self, img: torch.Tensor) -> torch.Tensor:
        """transforms.Resize((S,S)) -> ToTensor -> Normalize(ImageNet): uint8 [N][H][W][3] -> f32 (N,3,S,S)."""
        img = img.unsqueeze(0) if img.dim() == 3 else img
        r = self.resize(img, self.S, self.S, "bilinear")
        out = torch.empty(r.shape[0], 3, self.S, self.S, device=self.device)
        hip.u8_to_tensor(r, 0, 0, self.S, self.S, self.im_mean, self.im_std, out)
        return out

    def mask_input(self, gt: torch.Tensor) -> torch.Tensor:
        """datasets/wrappers.py:29-32 `mask_transform` without the float step: Resize((S, S), NEAREST) of a uint8 (h, w) /
        (N, h, w) ground-truth mask -> uint8 (N, S, S).  (`ToTensor` then divides by 255 and `calc_cod` multiplies it back and
        thresholds at 128, utils.py:155, sod_metric.py:21: the uint8 levels decide.)"""
        g = gt.unsqueeze(0) if gt.dim() == 2 else gt
        return self.resize(g.unsqueeze(-1), self.S, self.S, "nearest")[..., 0]

    def clip_input(self, img: torch.Tensor) -> torch.Tensor:
        """Resize(R, BICUBIC) -> CenterCrop(R) -> ToTensor -> Normalize(OpenAI) -> f32 (N,3,R,R)."""
        img = img.unsqueeze(0) if img.dim() == 3 else img
        N, H, W, _ = img.shape
        R = self.R
        rh, rw = (int(R * H / W), R) if W <= H else (R, int(R * W / H))
        r = self.resize(img, rh, rw, "bicubic")
        top, left = int(round((rh - R) / 2.0)), int(round((rw - R) / 2.0))
        out = torch.empty(N, 3, R, R, device=self.device)
        hip.u8_to_tensor(r, top, left, R, R, self.cl_mean, self.cl_std, out)
        return out

    def clip_mask(self, n: int, convention: str = "wrapper") -> torch.Tensor:
        """mask_transform on an all-ones mask: "wrapper" = (1 - 0.5) / 0.26 (uint8 mask, wrappers.py:62);
        "demo" = (255 - 0.5) / 0.26 (demo.py:103: float64 array, not rescaled by ToTensor)."""
        return torch.full((n, 1, self.R, self.R), clip_mask_value(convention), dtype=torch.float32, device=self.device)

    def __call__(self, img: torch.Tensor):
        if img.dim() == 3:
            img = img.unsqueeze(0)
        return self.sam_input(img), self.clip_input(img), self.clip_mask(img.shape[0])
