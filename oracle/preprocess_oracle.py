"""CPU ORACLE (test infrastructure only) for the step BEFORE the hot path: the
reference's image preprocessing

  SAM  input : transforms.Resize((S, S)) [PIL bilinear, antialiased] -> ToTensor -> Normalize(ImageNet)
               (demo.py:93-98, datasets/wrappers.py:22-27)
  CLIP input : Resize(n_px, BICUBIC) [shorter side] -> CenterCrop(n_px) -> ToTensor -> Normalize(OpenAI)
               (alpha_clip_rw/alpha_clip.py:79-86)

restated in numpy.  torchvision's Resize on a PIL image calls ``PIL.Image.resize`` -- third-party code that is
not in /root/reference (Pillow, libImaging/Resample.c; this container has Pillow 12.2).  Its published algorithm
is restated here (8-bit path: double-precision separable coefficients with support scaled by the shrink factor,
normalised, rounded to 22-bit fixed point; horizontal pass then vertical pass, each rounded to uint8) and pinned
bit-for-bit against Pillow itself by tests/test_preprocess.py and the golden file tests/golden/preprocess.npz.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
OPENAI_MEAN, OPENAI_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def _bilinear(x: float) -> float:
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def precompute_coeffs(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: (bounds (out,2) int32, kk (out,ksize) int32, ksize)."""
    f, fsupport = FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One separable pass on uint8 HWC along `axis` (1 = horizontal, 0 = vertical), rounded to uint8."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.int64)
    for xx in range(bounds.shape[0]):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_u8(img: np.ndarray, out_h: int, out_w: int, filt: str) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), filt) on a uint8 HWC image: horizontal pass, then vertical."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        b, k, _ = precompute_coeffs(w, out_w, filt)
        out = _pass(out, b, k, axis=1)
    if out_h != h:
        b, k, _ = precompute_coeffs(h, out_h, filt)
        out = _pass(out, b, k, axis=0)
    return out


def to_tensor_normalize(img_u8: np.ndarray, mean, std) -> np.ndarray:
    """ToTensor (uint8 HWC -> float32 CHW / 255) + Normalize ((x - mean) / std), all float32 like torch."""
    t = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


def clip_resize_shape(h: int, w: int, n_px: int) -> Tuple[int, int]:
    """torchvision Resize(int): shorter side -> n_px, other = int(n_px * long / short)."""
    if w <= h:
        return int(n_px * h / w), n_px
    return n_px, int(n_px * w / h)


def center_crop_box(h: int, w: int, n_px: int) -> Tuple[int, int]:
    return int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))


def sam_input(img_u8: np.ndarray, size: int) -> np.ndarray:
    return to_tensor_normalize(resize_u8(img_u8, size, size, "bilinear"), IMAGENET_MEAN, IMAGENET_STD)


def clip_input(img_u8: np.ndarray, n_px: int) -> np.ndarray:
    h, w = img_u8.shape[:2]
    rh, rw = clip_resize_shape(h, w, n_px)
    r = resize_u8(img_u8, rh, rw, "bicubic")
    top, left = center_crop_box(rh, rw, n_px)
    return to_tensor_normalize(r[top:top + n_px, left:left + n_px], OPENAI_MEAN, OPENAI_STD)
