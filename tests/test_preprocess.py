"""N1: preprocessing before the path.  CPU: the numpy restatement of Pillow's resample is pinned
bit-for-bit to golden vectors produced by Pillow (tools/make_preprocess_golden.py) and, when Pillow is importable, to
Pillow directly.  GPU: the HIP kernels must equal the oracle bit-for-bit (uint8) / exactly (fp32 normalise)."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P


@pytest.fixture(scope="module")
def gold(golden_dir):
    with np.load(os.path.join(golden_dir, "preprocess.npz")) as z:
        return {k: z[k] for k in z.files}


def _big(gold):
    rng = np.random.default_rng(int(gold["big_seed"][0]))
    for _, h, w, _, _, _ in [("a", 97, 131, 0, 0, 0), ("b", 60, 45, 0, 0, 0), ("c", 300, 200, 0, 0, 0),
                             ("d", 50, 70, 0, 0, 0), ("e", 90, 160, 0, 0, 0)]:
        rng.integers(0, 256, (h, w, 3), dtype=np.uint8)           # replay the generator's stream
    return rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)


def test_oracle_resample_matches_pillow_golden(gold):
    for name in "abcde":
        h, w, oh, ow, f = gold[f"meta_{name}"].tolist()
        got = P.resize_u8(gold[f"in_{name}"], oh, ow, "bicubic" if f else "bilinear")
        assert np.array_equal(got, gold[f"out_{name}"]), name


def test_host_coefficient_tables_equal_the_pillow_pinned_oracle():
    """the product's vectorised table builder (camouflaged_vlm_amd.preprocess._coeffs: all output indices at once, the window sum
    still left to right) against the oracle's per-index restatement of Pillow's precompute_coeffs / normalize_coeffs_8bpc:
    bounds and 22-bit fixed-point taps, bit for bit -- up- and down-scaling, sizes of 1-3 pixels, both filters"""
    from camouflaged_vlm_amd import preprocess as G
    rng = np.random.default_rng(0)
    cases = [(1920, 1024), (1080, 1024), (2048, 336), (1365, 336), (600, 1024), (1024, 1024), (337, 336), (5000, 336), (7, 1024),
             (1024, 7), (3, 2), (2, 3), (1, 5), (5, 1)] + [(int(a), int(b)) for a, b in rng.integers(1, 2500, (40, 2))]
    for n_in, n_out in cases:
        for filt in ("bilinear", "bicubic"):
            b, k = G._coeffs(n_in, n_out, filt)
            ob, ok, ks = P.precompute_coeffs(n_in, n_out, filt)
            assert k.shape == (n_out, ks) and b.dtype == np.int32 and k.dtype == np.int32
            assert np.array_equal(b, np.asarray(ob).reshape(b.shape)) and np.array_equal(k, np.asarray(ok).reshape(k.shape)), (n_in, n_out, filt)


def test_oracle_full_size_checksums(gold):
    big = _big(gold)
    assert zlib.crc32(big.tobytes()) == int(gold["big_crc"][0])
    assert zlib.crc32(P.resize_u8(big, 1024, 1024, "bilinear").tobytes()) == int(gold["big_crc"][1])
    assert zlib.crc32(P.resize_u8(big, 336, 448, "bicubic").tobytes()) == int(gold["big_crc"][2])


def test_oracle_against_live_pillow_if_present():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (123, 77, 3), dtype=np.uint8)
    for (oh, ow, f, pf) in [(64, 200, "bilinear", Image.BILINEAR), (336, 210, "bicubic", Image.BICUBIC)]:
        assert np.array_equal(P.resize_u8(a, oh, ow, f), np.asarray(Image.fromarray(a).resize((ow, oh), pf)))


def test_host_coefficient_tables_equal_oracle():
    from camouflaged_vlm_amd.preprocess import _coeffs
    for n_in, n_out, f in [(640, 1024, "bilinear"), (480, 336, "bicubic"), (97, 64, "bilinear"), (50, 120, "bicubic")]:
        b, k = _coeffs(n_in, n_out, f)
        b2, k2, _ = P.precompute_coeffs(n_in, n_out, f)
        assert np.array_equal(b, b2) and np.array_equal(k, k2)


def test_clip_geometry_helpers():
    assert P.clip_resize_shape(480, 640, 336) == (336, 448) and P.clip_resize_shape(640, 480, 336) == (448, 336)
    assert P.center_crop_box(336, 448, 336) == (0, 56)
    x = P.clip_input(np.full((50, 70, 3), 255, np.uint8), 32)
    assert x.shape == (3, 32, 32) and abs(float(x[0, 0, 0]) - (1 - 0.48145466) / 0.26862954) < 1e-6


@pytest.mark.gpu
def test_gpu_preprocess_bit_exact(gold):
    from camouflaged_vlm_amd.preprocess import GpuPreprocess
    pp = GpuPreprocess(inp_size=1024, clip_size=336)
    for name in "abcde":
        h, w, oh, ow, f = gold[f"meta_{name}"].tolist()
        img = torch.from_numpy(gold[f"in_{name}"]).cuda().unsqueeze(0)
        got = pp.resize(img, oh, ow, "bicubic" if f else "bilinear")[0].cpu().numpy()
        assert np.array_equal(got, gold[f"out_{name}"]), name
    big = _big(gold)
    dev = torch.from_numpy(big).cuda()
    inp, clip_image, clip_mask = pp(dev)
    torch.cuda.synchronize()
    assert inp.shape == (1, 3, 1024, 1024) and clip_image.shape == (1, 3, 336, 336) and clip_mask.shape == (1, 1, 336, 336)
    assert np.array_equal(inp[0].cpu().numpy(), P.sam_input(big, 1024))           # fp32: exact
    assert np.array_equal(clip_image[0].cpu().numpy(), P.clip_input(big, 336))
    assert abs(float(clip_mask[0, 0, 0, 0]) - 1.9230769) < 1e-6
    # batch of two equal-size images == two single calls
    two = torch.stack([dev, torch.flip(dev, dims=(1,))])
    b2 = pp.sam_input(two)
    assert torch.equal(b2[0], inp[0]) and torch.equal(b2[1], pp.sam_input(torch.flip(dev, dims=(1,)).contiguous())[0])


@pytest.mark.gpu
def test_gpu_nearest_mask_resize_equals_pillow():
    """datasets/wrappers.py:29-32 `mask_transform`: Resize((S, S), NEAREST) of the uint8 ground truth -- `GpuPreprocess.mask_input`
    against Pillow itself (and against its index rule, `preprocess.nearest_indices`, when Pillow is not importable)"""
    from camouflaged_vlm_amd.preprocess import GpuPreprocess, nearest_indices
    rng = np.random.default_rng(6)
    pre = GpuPreprocess(320, 56, "cuda")
    for h, w in ((97, 131), (683, 1024), (320, 320), (7, 5), (1365, 2048), (1024, 683), (600, 800)):
        a = rng.integers(0, 256, (h, w)).astype(np.uint8)
        got = pre.mask_input(torch.from_numpy(a).cuda())[0].cpu().numpy()
        want = a[nearest_indices(h, 320)][:, nearest_indices(w, 320)]
        try:
            from PIL import Image
            assert np.array_equal(want, np.asarray(Image.fromarray(a).resize((320, 320), Image.NEAREST)))
        except ImportError:
            pass
        assert np.array_equal(got, want), (h, w)


def test_nearest_index_rule_equals_pillow():
    """Pillow's NEAREST resize advances the source coordinate by repeated addition (ImagingScaleAffine): `nearest_indices` against
    Pillow itself on scales where (x + 0.5) * scale would land on the other side of an integer (1024 -> 320 = 3.2, ...)"""
    Image = pytest.importorskip("PIL.Image")
    from camouflaged_vlm_amd.preprocess import nearest_indices
    rng = np.random.default_rng(0)
    sizes = [(683, 1024, 320, 320), (1365, 2048, 1024, 1024), (1024, 683, 320, 320), (7, 5, 13, 11), (600, 800, 1024, 1024)]
    sizes += [tuple(int(v) for v in rng.integers(1, 1500, 4)) for _ in range(40)]
    for h, w, oh, ow in sizes:
        a = rng.integers(0, 256, (h, w)).astype(np.uint8)
        want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.NEAREST))
        assert np.array_equal(a[nearest_indices(h, oh)][:, nearest_indices(w, ow)], want), (h, w, oh, ow)
