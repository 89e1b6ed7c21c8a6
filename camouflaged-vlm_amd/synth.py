"""Deterministic synthetic weights and inputs (no network, no checkpoints, no dataset).

Counter-based: every tensor is drawn from a Philox stream keyed by sha256(name) and the seed, so a
tensor's values depend only on (name, shape, kind, seed) -- not on construction order, not on
torch's RNG.  The same tensors are therefore available in this container (loaded into the reference
by tools/make_golden.py) and on the GPU box (loaded into the HIP path).

Scales follow SURVEY.md §8(d): PyTorch-default-like fan-in scaling for linears/convs, *non-zero*
rel_pos / pos_embed / conv1_alpha (the reference zero-initialises them, image_encoder.py:78-80,
485-486; alpha_clip_rw/model.py:878-881), hyper-network tails scaled up so mask logits span
several units, logit_scale = ln(100).
"""
from __future__ import annotations

import hashlib
import os
import math
from typing import Dict, Iterable, Tuple

import numpy as np

from .spec import ClipGeometry, Entry, SamGeometry, full_entries


def _rng(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = np.frombuffer(h[:16], dtype=np.uint64).copy()
    return np.random.Generator(np.random.Philox(key=key))


def _normal(name: str, seed: int, shape: Tuple[int, ...]) -> np.ndarray:
    return _rng(name, seed).standard_normal(size=shape, dtype=np.float32)


def _uniform(name: str, seed: int, shape: Tuple[int, ...]) -> np.ndarray:
    return _rng(name, seed).random(size=shape, dtype=np.float32) * 2.0 - 1.0


def make_tensor(name: str, shape: Tuple[int, ...], kind: str, seed: int = 0) -> np.ndarray:
    if kind in ("linear", "hyper_tail"):
        fan_in = shape[1]
        w = _uniform(name, seed, shape) * (1.0 / math.sqrt(fan_in))
        # the mask-logit heads are scaled so logits have std of a few units
        return w * (24.0 if kind == "hyper_tail" else 1.0)
    if kind == "conv":
        fan_in = int(np.prod(shape[1:]))
        return _uniform(name, seed, shape) * (1.0 / math.sqrt(fan_in))
    if kind == "convT":
        # ConvTranspose2d weight is (in, out, kh, kw); torch's fan_in uses dim 1
        fan_in = int(shape[1] * shape[2] * shape[3])
        return _uniform(name, seed, shape) * (1.0 / math.sqrt(fan_in))
    if kind == "bias":
        return _uniform(name, seed, shape) * 0.05
    if kind == "ln_w":
        return 1.0 + 0.1 * _normal(name, seed, shape)
    if kind == "ln_b":
        return 0.05 * _normal(name, seed, shape)
    if kind == "relpos":
        return 0.1 * _normal(name, seed, shape)
    if kind == "pos":
        return 0.1 * _normal(name, seed, shape)
    if kind == "embed":
        return 0.05 * _normal(name, seed, shape)
    if kind == "embed_w":
        return (shape[-1] ** -0.5) * _normal(name, seed, shape)
    if kind == "proj_w":
        return (shape[0] ** -0.5) * _normal(name, seed, shape)
    if kind == "gauss":
        return _normal(name, seed, shape)
    if kind == "logit_scale":
        return np.asarray(math.log(100.0), dtype=np.float32)
    raise ValueError(f"unknown tensor kind {kind!r} for {name}")


def make_state_dict(entries: Iterable[Entry], seed: int = 0, threads: int = 0) -> Dict[str, np.ndarray]:
    """Every tensor has its own counter-based generator (keyed by its name), so the tensors can be drawn in any order and in
    parallel: `threads` worker threads (0: min(8, CPUs the process may use); numpy's generators release the GIL).  1.04 G parameters of
    the demo geometry take ~30 s on one thread -- times eight when eight ranks of one node build the same model."""
    entries = list(entries)
    if threads <= 0:
        try:
            threads = min(8, len(os.sched_getaffinity(0)))
        except AttributeError:
            threads = min(8, os.cpu_count() or 1)
    one = lambda e: np.ascontiguousarray(make_tensor(e[0], e[1], e[2], seed), dtype=np.float32)
    if threads == 1 or len(entries) < 8:
        return {e[0]: one(e) for e in entries}
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        vals = list(ex.map(one, entries))
    return {e[0]: v for e, v in zip(entries, vals)}


_FULL_CACHE: Dict[tuple, Dict[str, np.ndarray]] = {}


def make_full_state_dict(g: SamGeometry, c: ClipGeometry, seed: int = 0) -> Dict[str, np.ndarray]:
    """The assembled model's 1195 tensors.  The last two results are kept (a test session builds the demo geometry's 4 GB a dozen
    times): callers get a fresh dict of the SAME arrays and must copy before they modify one (apply_outliers does)."""
    key = (g, c, seed)
    if key not in _FULL_CACHE:
        while len(_FULL_CACHE) >= 2:
            _FULL_CACHE.pop(next(iter(_FULL_CACHE)))
        _FULL_CACHE[key] = make_state_dict(full_entries(g, c), seed)
        for arr in _FULL_CACHE[key].values():
            arr.setflags(write=False)                    # shared with every later caller: an aliasing in-place write must fail, not poison them
        # torch.from_numpy on a read-only array warns once per process that writes through the tensor are undefined: exactly the protection wanted
        import warnings
        warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
    return dict(_FULL_CACHE[key])


def make_text_bank(n_cls: int, dim: int, split: str, seed: int = 0) -> np.ndarray:
    """Unit-norm (n_cls, dim) stand-in for the reference's saved text banks
    (datasets/ovcamo_info/*CamoPromptsTextFeaturesViTB-14-336.pth: fp32, unit-norm rows)."""
    t = _normal(f"text_bank.{split}", seed, (n_cls, dim))
    return (t / np.linalg.norm(t, axis=-1, keepdims=True)).astype(np.float32)


def make_inputs(g: SamGeometry, c: ClipGeometry, batch: int, seed: int = 0, index0: int = 0):
    """Synthetic model inputs, one independent stream per image index so that image i of a batch
    equals image i generated alone (the reference is batch-size-1 only).

    inp        (B,3,S,S)  ~N(0,1) clipped to +-2.5   (ImageNet-normalised range, demo.py:93-98)
    clip_image (B,3,R,R)  ~N(0,1)                    (OpenAI-normalised range)
    clip_mask  (B,1,R,R)  = (1-0.5)/0.26             (alpha_clip.py:88-94 on an all-ones mask)
    """
    S, R = g.inp_size, c.image_resolution
    inp = np.empty((batch, 3, S, S), np.float32)
    clip_image = np.empty((batch, 3, R, R), np.float32)
    for b in range(batch):
        inp[b] = np.clip(_normal(f"input.inp.{index0 + b}", seed, (3, S, S)), -2.5, 2.5)
        clip_image[b] = _normal(f"input.clip_image.{index0 + b}", seed, (3, R, R))
    clip_mask = np.full((batch, 1, R, R), (1.0 - 0.5) / 0.26, np.float32)
    return inp, clip_image, clip_mask


def apply_outliers(sd: Dict[str, np.ndarray], prefix: str = "image_encoder.") -> Dict[str, np.ndarray]:
    """Copy of ``sd`` with a few SAM-encoder channels pushed far outside O(1), the way real ViT checkpoints look
    (tests/golden/tiny_outliers.npz is the reference's output on these weights):

      * residual stream: patch-embed output channels 3 and 77 x 1e3, channel 40 x 1e4  ("massive" channels that
        every block carries; LayerNorm then squeezes the ordinary channels to ~1e-2)
      * one dead channel: patch-embed output channel 11 x 1e-6
      * one hot MLP unit: block 1 lin1 row 5 x 1e4 (hidden activation ~1e4, lands on every residual channel
        through lin2), block 2 lin1 row 9 x 1e3
    """
    out = {k: v.copy() for k, v in sd.items()}
    pw, pb = prefix + "patch_embed.proj.weight", prefix + "patch_embed.proj.bias"
    for ch, f in ((3, 1e3), (77, 1e3), (40, 1e4), (11, 1e-6)):
        out[pw][ch] *= f
        out[pb][ch] *= f
    for blk, row, f in ((1, 5, 1e4), (2, 9, 1e3)):
        out[f"{prefix}blocks.{blk}.mlp.lin1.weight"][row] *= f
        out[f"{prefix}blocks.{blk}.mlp.lin1.bias"][row] *= f
    return out


def make_openai_clip_state_dict(c: ClipGeometry, seed: int = 0, vocab_size: int = 49408) -> Dict[str, np.ndarray]:
    """The synthetic CLIP weights under OpenAI's key names, the way a JIT archive's state_dict carries them
    (alpha_clip_rw/model.py:825-884 reads this layout): `visual.*` with packed `attn.in_proj_weight/bias`, bare
    `transformer.*` / `positional_embedding` / `text_projection` / `ln_final.*` for the text tower, `token_embedding.weight`,
    the three metadata entries -- and NO `visual.conv1_alpha.weight` (Alpha-CLIP adds it, zero-initialised)."""
    from .spec import clip_entries
    own = make_state_dict([e for e in clip_entries(c, prefix="") if not e[0].startswith("prompt_learner.")], seed)
    out: Dict[str, np.ndarray] = {}
    for k, v in own.items():
        if k == "image_encoder.conv1_alpha.weight":
            continue
        if k.startswith("image_encoder."):
            k = "visual." + k[len("image_encoder."):].replace("attn.in_proj.weight", "attn.in_proj_weight").replace(
                "attn.in_proj.bias", "attn.in_proj_bias")
        elif k.startswith("text_encoder."):
            k = k[len("text_encoder."):]
        out[k] = v
    out["token_embedding.weight"] = make_tensor("openai.token_embedding.weight", (vocab_size, c.text_width), "embed", seed)
    out["input_resolution"] = np.asarray(c.image_resolution, np.int64)
    out["context_length"] = np.asarray(c.context_length, np.int64)
    out["vocab_size"] = np.asarray(vocab_size, np.int64)
    return out


def n3_rest_state_dict(g: SamGeometry, c: ClipGeometry, seed: int = 0) -> Dict[str, np.ndarray]:
    """Everything of the full synthetic state_dict that does NOT come from the CLIP archive: the SAM side and the MaPLe
    prompt learner's learnable tensors (ctx, proj, compound prompts and their projections).  Loaded with strict=False
    on top of an archive-initialised model (N3 test, tools/make_golden.py --only-n3)."""
    sd = make_full_state_dict(g, c, seed)
    drop = ("clip_model.image_encoder.", "clip_model.text_encoder.", "clip_model.logit_scale",
            "clip_model.prompt_learner.token_")
    return {k: v for k, v in sd.items() if not k.startswith(drop)}
