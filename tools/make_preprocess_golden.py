#!/usr/bin/env python3
"""Golden vectors for the preprocessing step (N1), produced with Pillow itself (the third-party library
torchvision's Resize calls) on synthetic uint8 images: tests/golden/preprocess.npz."""
import os, sys, zlib
import numpy as np
from PIL import Image
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(1234)
out = {}
cases = [("a", 97, 131, 64, 64, "bilinear"), ("b", 60, 45, 128, 128, "bilinear"), ("c", 300, 200, 168, 112, "bicubic"),
         ("d", 50, 70, 120, 90, "bicubic"), ("e", 90, 160, 56, 99, "bicubic")]
for name, h, w, oh, ow, f in cases:
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    r = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR if f == "bilinear" else Image.BICUBIC))
    out[f"in_{name}"], out[f"out_{name}"] = a, r
    out[f"meta_{name}"] = np.array([h, w, oh, ow, 0 if f == "bilinear" else 1])
# full-size case: checksums only (480x640 -> 1024^2 bilinear, -> 336x448 bicubic)
big = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
sam = np.asarray(Image.fromarray(big).resize((1024, 1024), Image.BILINEAR))
clip = np.asarray(Image.fromarray(big).resize((448, 336), Image.BICUBIC))
out["big_seed"] = np.array([1234])
out["big_crc"] = np.array([zlib.crc32(big.tobytes()), zlib.crc32(sam.tobytes()), zlib.crc32(clip.tobytes())], dtype=np.int64)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "preprocess.npz"), **out)
print("written", {k: v.shape for k, v in out.items() if k.startswith("out_")}, out["big_crc"])
