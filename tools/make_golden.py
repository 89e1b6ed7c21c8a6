#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself on CPU (build container only).

The reference tree (/root/reference) is imported read-only with container-local stubs for the
third-party packages that are absent here (SURVEY.md §8c recipe); nothing of it is copied.  The
synthetic weights/inputs come from this repo's own generator (camouflaged_vlm_amd.synth), are loaded
into the reference modules with ``load_state_dict(strict=True)`` (which is also the check that
spec.py restates the reference's key layout exactly), and the reference's outputs are saved as
small ``.npz`` fixtures under tests/golden/.

Usage:  python tools/make_golden.py [--out tests/golden] [--demo-digest] [--only-tiny] [--only-train-branch] [--only-demo-digest|--only-demo-alpha-digest|--only-hires-digest [--images N] [--check]]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CVLM_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402

import camouflaged_vlm_amd as cv  # noqa: E402
from camouflaged_vlm_amd import spec, synth  # noqa: E402


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_reference():
    """Make ``models.*``, ``cocotrainers.mapleAlphaCLIP`` and ``alpha_clip_rw`` of the reference importable."""
    _stub("loralib")
    _stub("dassl")
    _stub("dassl.engine", TrainerX=object)
    _stub("dassl.utils", load_checkpoint=None, load_pretrained_weights=None)
    _stub("dassl.optim", build_optimizer=None, build_lr_scheduler=None)
    _stub("ftfy", fix_text=lambda s: s)
    ph = type("P", (), {"__init__": lambda self, *a, **k: None})
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Compose=ph, Resize=ph, CenterCrop=ph, ToTensor=ph,
                          Normalize=ph, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    _stub("utils", log=lambda *a, **k: None)
    # parent packages whose __init__ pulls mmcv/open_clip: register bare packages with the right __path__
    for name, rel in (("models", "models"), ("models.mmseg", "models/mmseg"),
                      ("models.mmseg.models", "models/mmseg/models")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = m
    sys.path.insert(0, REF)
    import importlib
    mm = importlib.import_module("models.models")
    sys.modules["models"].register = mm.register
    sys.modules["models"].make = mm.make
    # bank files were saved from CUDA; the wrapper passes no map_location (sam_maskdecoder_edge.py:177-182)
    _orig = torch.load
    torch.load = lambda f, *a, **k: _orig(f, *a, **{**k, "map_location": "cpu", "weights_only": True})
    importlib.import_module("models.sam_maskdecoder_edge")
    ml = importlib.import_module("cocotrainers.mapleAlphaCLIP")
    cm = importlib.import_module("alpha_clip_rw.model")
    ns = {}
    with open(os.path.join(REF, "datasets/ovcamo_info/class_names.py")) as f:
        exec(f.read(), ns)
    return mm, ml, cm, ns["TRAIN_CLASS_NAMES"], ns["TEST_CLASS_NAMES"]


DEMO_ALPHA = (255.0 - 0.5) / 0.26          # demo.py:102-104 through alpha_clip.py:88-94 (Normalize(0.5, 0.26), no /255)


class DotDict:
    def __init__(self, d):
        for k, v in d.items():
            setattr(self, k, DotDict(v) if isinstance(v, dict) else v)


def build_reference(mm, ml, cm, g: spec.SamGeometry, c: spec.ClipGeometry, train_names, test_names, seed=0):
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        clip = cm.CLIP(c.embed_dim, c.image_resolution, c.vision_layers, c.vision_width, c.patch_size,
                       c.context_length, 49408, c.text_width, c.text_heads, c.text_layers,
                       design_details={"trainer": "MaPLe", "vision_depth": 0, "language_depth": 0,
                                       "vision_ctx": 0, "language_ctx": 0, "maple_length": c.n_ctx})
        cfg = DotDict({"MODEL": {"BACKBONE": {"NAME": "ViT-L/14@336px"}},
                       "TRAINER": {"MAPLE": {"N_CTX": c.n_ctx, "CTX_INIT": "a photo of a", "PREC": "fp32",
                                             "PROMPT_DEPTH": c.prompt_depth}},
                       "INPUT": {"SIZE": [c.image_resolution, c.image_resolution]}})
        custom = ml.CustomCLIP(cfg, train_names[:c.n_cls_train], test_names[:c.n_cls_test], clip.float())
        enc = dict(name="sam", img_size=g.inp_size, mlp_ratio=g.mlp_ratio, patch_size=g.patch_size,
                   qkv_bias=True, use_rel_pos=True, window_size=g.window_size, out_chans=g.out_chans,
                   scale_factor=32, input_type="fft", freq_nums=0.25, prompt_type="highpass",
                   prompt_embed_dim=g.prompt_embed_dim, tuning_stage=1234, handcrafted_tune=True,
                   embedding_tune=True, adaptor="adaptor", embed_dim=g.embed_dim, depth=g.depth,
                   num_heads=g.num_heads, global_attn_indexes=list(g.global_attn_indexes))
        model = mm.make({"name": "sam_maskdecoder_edge", "args": {"inp_size": g.inp_size, "loss": "iou",
                                                                 "encoder_mode": enc}})
        model.device = torch.device("cpu")
        # use the first n rows of the reference's real banks
        model.train_text_features = model.train_text_features[:c.n_cls_train].float()
        model.test_text_features = model.test_text_features[:c.n_cls_test].float()
        model.load_mapleAlphaCLIP(custom)
    finally:
        os.chdir(cwd)
    sd = synth.make_full_state_dict(g, c, seed)
    ref_keys = set(model.state_dict().keys())
    assert ref_keys == set(sd.keys()), (sorted(ref_keys - set(sd))[:8], sorted(set(sd) - ref_keys)[:8])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    eot_test = custom.tokenized_prompts_test.argmax(dim=-1).numpy().astype(np.int32)
    eot_train = custom.tokenized_prompts.argmax(dim=-1).numpy().astype(np.int32)
    return model, sd, eot_train, eot_test


def run_reference(model, inp, clip_image, clip_mask, R):
    """demo.py:110-122, one image at a time (the reference is B=1 only)."""
    import torch.nn.functional as F
    masks, preds, logits, l1 = [], [], [], []
    with torch.no_grad():
        for b in range(inp.shape[0]):
            i, ci, cm_ = (torch.from_numpy(t[b:b + 1]) for t in (inp, clip_image, clip_mask))
            m = model.infer_test(i, ci, cm_)
            _, _, _, s1 = model.clip_model(ci, cm_, train=False)
            alpha = F.interpolate(torch.sigmoid(m), (R, R), mode="bilinear", align_corners=False)
            _, _, p, s = model.clip_model(ci, alpha, train=False)
            masks.append(m.numpy()); preds.append(p.numpy()); logits.append(s.numpy()); l1.append(s1.numpy())
    return np.concatenate(masks), np.concatenate(preds), np.concatenate(logits), np.concatenate(l1)


def tiny(out_dir, mods):
    mm, ml, cm, train_names, test_names = mods
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    model, sd, eot_train, eot_test = build_reference(mm, ml, cm, g, c, train_names, test_names)
    inp, clip_image, clip_mask = synth.make_inputs(g, c, batch=2)
    taps = {}
    hooks = []

    def tap(name, mod, fn=lambda o: o):
        def hook(m, i, o):
            if name not in taps:            # first image only
                taps[name] = fn(o).detach().numpy().copy()
        hooks.append(mod.register_forward_hook(hook))

    enc = model.image_encoder
    tap("patch_embed", enc.patch_embed)
    for i, blk in enumerate(enc.blocks):
        tap(f"block{i}", blk)
    tap("attn0", enc.blocks[0].attn)           # windowed attention incl. pad rows (4 windows x 14 x 14)
    tap("attn1", enc.blocks[1].attn)           # global attention
    tap("features", enc, lambda o: o[0])
    tap("upscaled", model.mask_decoder.output_upscaling)
    tap("edge_features", model.mask_decoder.embedding_encoder)
    tap("hs", model.mask_decoder.transformer, lambda o: o[0])
    tap("src", model.mask_decoder.transformer, lambda o: o[1])
    tap("low_res_masks", model.mask_decoder, lambda o: o[0])
    # mask_decoder_edge.py:170,172-178: edge_embedding = embedding_maskfeature(upscaled) + edge_features (the sum is formed
    # below from the two module outputs); hypernetwork rows actually used by mask 0: mask token 0 and the edge token
    tap("maskfeature", model.mask_decoder.embedding_maskfeature)
    tap("hyper_mask0", model.mask_decoder.output_hypernetworks_mlps[0])
    tap("hyper_edge", model.mask_decoder.edge_mlp)
    tap("clip_visual", model.clip_model.image_encoder)
    tap("clip_text", model.clip_model.text_encoder)
    with torch.no_grad():
        hp = enc.prompt_generator.fft(torch.from_numpy(inp[:1]), 0.25).numpy()
        pe = model.get_dense_pe().numpy()
    masks, preds, logits, logits1 = run_reference(model, inp, clip_image, clip_mask, c.image_resolution)
    for h in hooks:
        h.remove()
    taps["edge_emb"] = taps.pop("maskfeature") + taps["edge_features"]
    # demo.py:102-104 feeds pass 1 an alpha of (255 - 0.5) / 0.26 = 978.8 (a float64 all-255 PIL image is not rescaled by
    # ToTensor): images 2 and 3 of the synthetic stream run with that value
    inp_d, clip_image_d, _ = synth.make_inputs(g, c, batch=2, index0=2)
    clip_mask_d = np.full((2, 1, c.image_resolution, c.image_resolution), DEMO_ALPHA, np.float32)
    masks_d, preds_d, logits_d, logits1_d = run_reference(model, inp_d, clip_image_d, clip_mask_d, c.image_resolution)
    np.savez_compressed(
        os.path.join(out_dir, "tiny_cascade.npz"),
        demoalpha_mask_logits=masks_d.astype(np.float32), demoalpha_pred=preds_d.astype(np.int64),
        demoalpha_class_logits=logits_d.astype(np.float32), demoalpha_pass1_logits=logits1_d.astype(np.float32),
        demoalpha_value=np.float32(DEMO_ALPHA),
        mask_logits=masks.astype(np.float32), pred=preds.astype(np.int64), class_logits=logits.astype(np.float32),
        pass1_logits=logits1.astype(np.float32), eot_train=eot_train, eot_test=eot_test,
        bank_test=model.test_text_features.numpy(), bank_train=model.train_text_features.numpy(),
        highpass=hp.astype(np.float32), dense_pe=pe.astype(np.float32),
        **{"tap_" + k: v.astype(np.float32) for k, v in taps.items()})
    print("tiny: mask std %.3f min %.3f max %.3f | pred %s | logits[0,:3] %s" %
          (masks.std(), masks.min(), masks.max(), preds, logits[0, :3]))
    frac = float((np.abs(masks) < 1e-3).mean())
    print("tiny: frac |logit|<1e-3 = %.2e ; positive frac %.3f" % (frac, float((masks > 0).mean())))


def tokens(out_dir, mods):
    """G4 constants: tokenised OVCamo prompts (mapleAlphaCLIP.py:132-168) and the real text banks."""
    mm, ml, cm, train_names, test_names = mods
    import importlib
    ac = importlib.import_module("alpha_clip_rw.alpha_clip")
    fmt = lambda names: ["a photo of a " + n.replace("_", " ") + "." for n in names]
    tk_train = torch.cat([ac.tokenize(p) for p in fmt(train_names)]).numpy().astype(np.int32)
    tk_test = torch.cat([ac.tokenize(p) for p in fmt(test_names)]).numpy().astype(np.int32)
    banks = {}
    for split in ("Train", "Test"):
        banks[split.lower()] = torch.load(
            os.path.join(REF, f"datasets/ovcamo_info/{split}CamoPromptsTextFeaturesViTB-14-336.pth")).float().numpy()
    np.savez_compressed(os.path.join(out_dir, "ovcamo_constants.npz"), tokens_train=tk_train, tokens_test=tk_test,
                        bank_train=banks["train"], bank_test=banks["test"],
                        names_train=np.array(train_names), names_test=np.array(test_names))
    print("tokens:", tk_train.shape, tk_test.shape, "eot test", tk_test.argmax(-1)[:8])


N_DENSE, N_NEAR = 65536, 4096        # round 6: mask logits kept per image -- a dense common sample and each image's pixels nearest the decision


def demo_digest(out_dir, mods, n_images=16, check=False, seed=0, outliers=False, name="demo_digest.npz"):
    """G3: full demo.yaml geometry, B=1 per image (the reference is B=1 only), digests only.  Every array is PER IMAGE
    (leading dimension n_images): packed mask bits, 4096 sampled mask logits (the same positions for every image),
    class logits of both CLIP passes, predictions.  Images are `synth.make_inputs(..., batch=n)` rows 0..n-1, i.e. image i
    does not depend on n: a run with fewer images reproduces a prefix of the committed file (--check compares that prefix).
    bench.py times batches of 8 (images 0-7 and 8-15 alternate; rank 1 of a multi-GPU run owns 8-15).

    Round 6: beside the 4096 positions of rounds 1-5 every image also keeps N_DENSE = 65536 common positions
    (`dense_idx` / `dense_samples`: 6 % of the pixels) and ITS OWN N_NEAR = 4096 positions of smallest |logit| (`near_idx` /
    `near_samples`: where a sign can flip); `seed` draws other synthetic weights (`demo_digest_seed1.npz`), `outliers` applies
    `synth.apply_outliers` to them (`demo_digest_outliers.npz`: massive residual channels, a hot MLP unit) -- the mx arithmetic's
    parity evidence no longer rests on one weight draw."""
    mm, ml, cm, train_names, test_names = mods
    g, c = spec.DEMO_SAM, spec.DEMO_CLIP
    model, sd, eot_train, eot_test = build_reference(mm, ml, cm, g, c, train_names, test_names, seed=seed)
    if outliers:
        sd = outlier_weights(sd)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    inp, clip_image, clip_mask = synth.make_inputs(g, c, batch=n_images)
    import time
    rng = np.random.default_rng(0)
    idx = rng.integers(0, inp.shape[2] * inp.shape[3], size=4096)
    dense_idx = np.random.default_rng(6).choice(inp.shape[2] * inp.shape[3], size=N_DENSE, replace=False).astype(np.int32)
    dense, near_i, near_v = [], [], []
    bits, samples, stats, preds, logits, logits1, secs = [], [], [], [], [], [], []
    # encoder output (B,256,64,64) of the same run: the `--workload encoder` bench line (BASELINE configs[1]) is checked
    # against these samples / channel means; the first infer_test call of run_reference is the one recorded
    feat_idx = np.random.default_rng(2).integers(0, g.out_chans * g.grid * g.grid, size=16384)
    feats, fsamples, fcmean = [], [], []
    hook = model.image_encoder.register_forward_hook(
        lambda m_, i_, o_: feats.append((o_[0] if isinstance(o_, (tuple, list)) else o_).detach().numpy().copy()))
    for b in range(n_images):
        t0 = time.time()
        feats.clear()
        m, p, s, s1 = run_reference(model, inp[b:b + 1], clip_image[b:b + 1], clip_mask[b:b + 1], c.image_resolution)
        secs.append(time.time() - t0)
        fsamples.append(feats[0].reshape(-1)[feat_idx].astype(np.float32))
        fcmean.append(feats[0].mean(axis=(0, 2, 3)).astype(np.float32))
        bits.append(np.packbits(m > 0)); samples.append(m.reshape(-1)[idx])
        flat = m.reshape(-1)
        dense.append(flat[dense_idx])
        ni = np.sort(np.argpartition(np.abs(flat), N_NEAR)[:N_NEAR]).astype(np.int32)
        near_i.append(ni); near_v.append(flat[ni])
        stats.append([m.mean(), m.std(), m.min(), m.max()])
        preds.append(p[0]); logits.append(s[0]); logits1.append(s1[0])
        print("demo: image %d in %.1f s; mask std %.3f; pred %s; largest |logit| of the near set %.3e" % (b, secs[-1], m.std(), p, np.abs(flat[ni]).max()), flush=True)
    out = dict(mask_bits=np.stack(bits), mask_samples=np.stack(samples).astype(np.float32), sample_idx=idx,
               mask_stats=np.array(stats, np.float64), pred=np.array(preds, np.int64),
               class_logits=np.stack(logits).astype(np.float32), pass1_logits=np.stack(logits1).astype(np.float32),
               eot_test=eot_test, eot_train=eot_train, feat_idx=feat_idx, feat_samples=np.stack(fsamples),
               feat_channel_mean=np.stack(fcmean), dense_idx=dense_idx, dense_samples=np.stack(dense).astype(np.float32),
               near_idx=np.stack(near_i), near_samples=np.stack(near_v).astype(np.float32))
    if seed or outliers:
        out.update(weight_seed=np.int64(seed), outlier_weights=np.bool_(outliers))
    hook.remove()
    path = os.path.join(out_dir, name)
    if check:
        _check_prefix(path, out, n_images)
        return
    np.savez_compressed(path, ref_seconds=np.array(secs), threads=np.array(torch.get_num_threads()), **out)


def demo_alpha_digest(out_dir, mods, n_images=1, check=False):
    """BASELINE configs[0] IS demo.py: the full demo.yaml geometry with demo.py's own pass-1 alpha, (255 - 0.5) / 0.26 = 978.8
    (demo.py:102-104), on images 0..n-1 of the synthetic stream (the inputs of demo_digest.npz; only the
    alpha differs).  Same per-image arrays as `demo_digest`."""
    mm, ml, cm, train_names, test_names = mods
    g, c = spec.DEMO_SAM, spec.DEMO_CLIP
    model, sd, eot_train, eot_test = build_reference(mm, ml, cm, g, c, train_names, test_names)
    inp, clip_image, _ = synth.make_inputs(g, c, batch=n_images)
    clip_mask = np.full((n_images, 1, c.image_resolution, c.image_resolution), DEMO_ALPHA, np.float32)
    idx = np.random.default_rng(0).integers(0, inp.shape[2] * inp.shape[3], size=4096)
    bits, samples, preds, logits, logits1 = [], [], [], [], []
    for b in range(n_images):
        m, p, s, s1 = run_reference(model, inp[b:b + 1], clip_image[b:b + 1], clip_mask[b:b + 1], c.image_resolution)
        bits.append(np.packbits(m > 0)); samples.append(m.reshape(-1)[idx])
        preds.append(p[0]); logits.append(s[0]); logits1.append(s1[0])
        print("demo alpha: image %d; mask std %.3f; pred %s; pass-1 logits[:3] %s" % (b, m.std(), p, s1[0, :3]), flush=True)
    out = dict(mask_bits=np.stack(bits), mask_samples=np.stack(samples).astype(np.float32), sample_idx=idx,
               pred=np.array(preds, np.int64), class_logits=np.stack(logits).astype(np.float32),
               pass1_logits=np.stack(logits1).astype(np.float32), eot_test=eot_test, alpha=np.float32(DEMO_ALPHA))
    path = os.path.join(out_dir, "demo_alpha_digest.npz")
    if check:
        _check_prefix(path, out, n_images)
        return
    np.savez_compressed(path, **out)


def train_branch(out_dir, mods, check=False):
    """`CustomCLIP.forward(image, mask, train=True)` (cocotrainers/mapleAlphaCLIP.py:267-280): the same towers on the 14 TRAIN prompts
    and the train bank -- forward-only arithmetic (VERDICT r4 missing #1).  Tiny geometry: two images; full demo geometry: image 0 with
    the dataset wrapper's alpha (1.923)."""
    mm, ml, cm, train_names, test_names = mods
    out = {}
    for tag, (g, c), n in (("tiny", (spec.TINY_SAM, spec.TINY_CLIP), 2), ("demo", (spec.DEMO_SAM, spec.DEMO_CLIP), 1)):
        model, sd, eot_train, eot_test = build_reference(mm, ml, cm, g, c, train_names, test_names)
        _, clip_image, clip_mask = synth.make_inputs(g, c, batch=n)
        with torch.no_grad():
            img, sel, pred, logits = model.clip_model(torch.from_numpy(clip_image), torch.from_numpy(clip_mask), train=True)
        assert logits.shape == (n, c.n_cls_train)
        out.update({f"{tag}_img": img.numpy().astype(np.float32), f"{tag}_sel": sel.numpy().astype(np.float32),
                    f"{tag}_pred": pred.numpy().astype(np.int64), f"{tag}_logits": logits.numpy().astype(np.float32),
                    f"{tag}_eot_train": eot_train, f"{tag}_bank_train": model.train_text_features.numpy().astype(np.float32)})
        print("train branch %s: logits %s pred %s" % (tag, logits.shape, pred.tolist()), flush=True)
        del model
    path = os.path.join(out_dir, "train_branch.npz")
    if check:
        _check_prefix(path, out, 0)
        return
    np.savez_compressed(path, **out)


def _check_prefix(path, out, n):
    """--check: the arrays just produced by the reference equal the first n images of the committed file, bit for bit."""
    with np.load(path) as z:
        worst = 0.0
        for k, v in out.items():
            ref = z[k]
            if ref.shape != v.shape:                     # per-image arrays: compare the prefix
                ref = ref[:n]
            d = float(np.abs(ref.astype(np.float64) - v.astype(np.float64)).max()) if v.size else 0.0
            worst = max(worst, d)
            print("check %-14s %-18s max abs diff %g" % (k, v.shape, d))
    print("check: %s (first %d image(s)) -> %s" % (os.path.basename(path), n, "IDENTICAL" if worst == 0.0 else "DIFFERENT"))
    if worst != 0.0:
        sys.exit(1)


def outlier_weights(sd):
    """Round 2, VERDICT item 6: the synthetic weights with a few channels pushed far outside O(1) -- what real ViT
    checkpoints look like (massive residual channels, one hot MLP unit, one dead channel).  Pure function of the
    synthetic state_dict so that the GPU test rebuilds exactly the same weights (camouflaged_vlm_amd.synth)."""
    return synth.apply_outliers(sd)


def tiny_outliers(out_dir, mods):
    """Tiny cascade on the outlier weights: masks / logits of the reference for tests/test_cascade_gpu.py."""
    mm, ml, cm, train_names, test_names = mods
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    model, sd, eot_train, eot_test = build_reference(mm, ml, cm, g, c, train_names, test_names)
    sd2 = outlier_weights(sd)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=True)
    inp, clip_image, clip_mask = synth.make_inputs(g, c, batch=2)
    taps = {}
    hooks = []
    enc = model.image_encoder
    for name, mod, fn in (("patch_embed", enc.patch_embed, lambda o: o), ("block0", enc.blocks[0], lambda o: o),
                          ("block3", enc.blocks[3], lambda o: o), ("features", enc, lambda o: o[0])):
        def hook(m, i, o, name=name, fn=fn):
            if name not in taps:
                taps[name] = fn(o).detach().numpy().copy()
        hooks.append(mod.register_forward_hook(hook))
    masks, preds, logits, logits1 = run_reference(model, inp, clip_image, clip_mask, c.image_resolution)
    for h in hooks:
        h.remove()
    np.savez_compressed(
        os.path.join(out_dir, "tiny_outliers.npz"),
        mask_logits=masks.astype(np.float32), pred=preds.astype(np.int64), class_logits=logits.astype(np.float32),
        pass1_logits=logits1.astype(np.float32), eot_test=eot_test, bank_test=model.test_text_features.numpy(),
        **{"tap_" + k: v.astype(np.float32) for k, v in taps.items()})
    print("outliers: |patch_embed| max %.3e, |block0| max %.3e, |block3| max %.3e; mask std %.3f min %.3f max %.3f | pred %s" %
          (np.abs(taps["patch_embed"]).max(), np.abs(taps["block0"]).max(), np.abs(taps["block3"]).max(),
           masks.std(), masks.min(), masks.max(), preds))


def hires_digest(out_dir, mods, n_images=4, check=False):
    """BASELINE configs[4] at its stated size: the reference's ImageEncoderViT *built* at 1536^2 with ViT-H width
    (96x96 tokens, S = 9216 global attention, 191-row rel-pos tables); per image (B = 1 forwards of images 0..n-1, the
    batch bench.py times at --geometry hires1536 --batch 4): 16384 samples of the (1,256,96,96) output + channel means."""
    import dataclasses
    import importlib
    import time
    from functools import partial
    g = dataclasses.replace(spec.DEMO_SAM, inp_size=1536)
    ie = importlib.import_module("models.mmseg.models.sam.image_encoder")
    enc = ie.ImageEncoderViT(depth=g.depth, embed_dim=g.embed_dim, img_size=g.inp_size, mlp_ratio=g.mlp_ratio,
                             norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=g.num_heads,
                             patch_size=g.patch_size, qkv_bias=True, use_rel_pos=True,
                             global_attn_indexes=list(g.global_attn_indexes), window_size=g.window_size,
                             out_chans=g.out_chans)
    sd = synth.make_state_dict(spec.sam_encoder_entries(g))
    ref_keys = set(enc.state_dict().keys())
    mine = {k[len("image_encoder."):]: v for k, v in sd.items()}
    assert ref_keys == set(mine.keys()), (sorted(ref_keys - set(mine))[:8], sorted(set(mine) - ref_keys)[:8])
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in mine.items()}, strict=True)
    enc.eval()
    inp = synth.make_inputs(g, spec.DEMO_CLIP, batch=n_images)[0]
    rng = np.random.default_rng(1)
    idx = None
    samples, stats, cmean, secs, shape = [], [], [], [], None
    for b in range(n_images):
        t0 = time.time()
        with torch.no_grad():
            o = enc(torch.from_numpy(inp[b:b + 1]))
        o = (o[0] if isinstance(o, (tuple, list)) else o).numpy()
        secs.append(time.time() - t0)
        if idx is None:
            idx, shape = rng.integers(0, o.size, size=16384), o.shape
        samples.append(o.reshape(-1)[idx].astype(np.float32))
        stats.append([o.mean(), o.std(), o.min(), o.max()])
        cmean.append(o.mean(axis=(0, 2, 3)).astype(np.float32))
        print("hires1536: image %d, encoder output %s in %.1f s; std %.4f min %.3f max %.3f" %
              (b, o.shape, secs[-1], o.std(), o.min(), o.max()), flush=True)
    out = dict(shape=np.array(shape), sample_idx=idx, samples=np.stack(samples), stats=np.array(stats, np.float64),
               channel_mean=np.stack(cmean))
    path = os.path.join(out_dir, "hires1536_digest.npz")
    if check:
        _check_prefix(path, out, n_images)
        return
    np.savez_compressed(path, ref_seconds=np.array(secs), **out)


def n3_openai(out_dir, mods):
    """N3: the reference's own loading path -- build_model(OpenAI-named state_dict) -> CustomCLIP (token vectors from the
    embedding table and the tokenizer) -> load_mapleAlphaCLIP -> strict=False load of the rest -- on a synthetic archive
    (synth.make_openai_clip_state_dict).  Saved: per-key checksums of the resulting CustomCLIP state_dict, the token
    buffers, and the cascade's outputs on two images."""
    mm, ml, cm, train_names, test_names = mods
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    osd = synth.make_openai_clip_state_dict(c)
    clip = cm.build_model({k: torch.from_numpy(np.asarray(v)) for k, v in osd.items()},
                          design_details={"trainer": "MaPLe", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0,
                                          "language_ctx": 0, "maple_length": c.n_ctx}).float()
    cfg = DotDict({"MODEL": {"BACKBONE": {"NAME": "ViT-L/14@336px"}},
                   "TRAINER": {"MAPLE": {"N_CTX": c.n_ctx, "CTX_INIT": "a photo of a", "PREC": "fp32",
                                         "PROMPT_DEPTH": c.prompt_depth}},
                   "INPUT": {"SIZE": [c.image_resolution, c.image_resolution]}})
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        custom = ml.CustomCLIP(cfg, train_names[:c.n_cls_train], test_names[:c.n_cls_test], clip)
        enc = dict(name="sam", img_size=g.inp_size, mlp_ratio=g.mlp_ratio, patch_size=g.patch_size,
                   qkv_bias=True, use_rel_pos=True, window_size=g.window_size, out_chans=g.out_chans,
                   scale_factor=32, input_type="fft", freq_nums=0.25, prompt_type="highpass",
                   prompt_embed_dim=g.prompt_embed_dim, tuning_stage=1234, handcrafted_tune=True,
                   embedding_tune=True, adaptor="adaptor", embed_dim=g.embed_dim, depth=g.depth,
                   num_heads=g.num_heads, global_attn_indexes=list(g.global_attn_indexes))
        model = mm.make({"name": "sam_maskdecoder_edge", "args": {"inp_size": g.inp_size, "loss": "iou", "encoder_mode": enc}})
        model.device = torch.device("cpu")
        model.train_text_features = model.train_text_features[:c.n_cls_train].float()
        model.test_text_features = model.test_text_features[:c.n_cls_test].float()
        model.load_mapleAlphaCLIP(custom)
    finally:
        os.chdir(cwd)
    rest = {k: torch.from_numpy(v) for k, v in synth.n3_rest_state_dict(g, c).items()}
    missing, unexpected = model.load_state_dict(rest, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("clip_model.") for k in missing), missing[:5]
    model.eval()
    inp, clip_image, clip_mask = synth.make_inputs(g, c, batch=2)
    masks, preds, logits, logits1 = run_reference(model, inp, clip_image, clip_mask, c.image_resolution)
    csd = custom.state_dict()
    keys = sorted(csd.keys())
    sums = np.array([[float(csd[k].double().sum()), float(csd[k].double().pow(2).sum())] for k in keys], np.float64)
    pl = "prompt_learner."
    np.savez_compressed(
        os.path.join(out_dir, "n3_openai_load.npz"), keys=np.array(keys), sums=sums,
        token_prefix=csd[pl + "token_prefix"].numpy(), token_suffix=csd[pl + "token_suffix"].numpy(),
        token_prefix_test=csd[pl + "token_prefix_test"].numpy(), token_suffix_test=csd[pl + "token_suffix_test"].numpy(),
        eot_test=custom.tokenized_prompts_test.argmax(dim=-1).numpy().astype(np.int32),
        eot_train=custom.tokenized_prompts.argmax(dim=-1).numpy().astype(np.int32),
        bank_test=model.test_text_features.numpy(), mask_logits=masks.astype(np.float32), pred=preds.astype(np.int64),
        class_logits=logits.astype(np.float32), pass1_logits=logits1.astype(np.float32))
    print("n3: %d CLIP keys; mask std %.3f; pred %s; conv1_alpha |max| %.1e" %
          (len(keys), masks.std(), preds, float(csd["image_encoder.conv1_alpha.weight"].abs().max())))


def sam_plain(out_dir, mods):
    """Registry entry ``sam`` (models/sam.py:298-440): encoder + vanilla MaskDecoder, no prompts -> `infer` masks."""
    import importlib
    mm = mods[0]
    _stub("open_clip")
    importlib.import_module("models.sam")
    g = spec.TINY_SAM
    enc = dict(name="sam", img_size=g.inp_size, mlp_ratio=g.mlp_ratio, patch_size=g.patch_size,
               qkv_bias=True, use_rel_pos=True, window_size=g.window_size, out_chans=g.out_chans,
               scale_factor=32, input_type="fft", freq_nums=0.25, prompt_type="highpass",
               prompt_embed_dim=g.prompt_embed_dim, tuning_stage=1234, handcrafted_tune=True,
               embedding_tune=True, adaptor="adaptor", embed_dim=g.embed_dim, depth=g.depth,
               num_heads=g.num_heads, global_attn_indexes=list(g.global_attn_indexes))
    model = mm.make({"name": "sam", "args": {"inp_size": g.inp_size, "loss": "iou", "encoder_mode": enc}})
    sd = synth.make_state_dict(spec.sam_plain_entries(g), 0)
    ref_keys = set(model.state_dict().keys())
    assert ref_keys == set(sd.keys()), (sorted(ref_keys - set(sd))[:8], sorted(set(sd) - ref_keys)[:8])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    inp, _, _ = synth.make_inputs(g, spec.TINY_CLIP, batch=2)
    taps = {}
    def tap(name, pick):
        def hook(m, i, o):
            if name not in taps:                # first image only
                taps[name] = pick(o).detach().numpy().copy()
        return hook
    h1 = model.mask_decoder.transformer.register_forward_hook(tap("hs", lambda o: o[0]))
    h2 = model.mask_decoder.register_forward_hook(tap("low_res_masks", lambda o: o[0]))
    with torch.no_grad():
        masks = np.concatenate([model.infer(torch.from_numpy(inp[b:b + 1])).numpy() for b in range(2)])
    h1.remove(); h2.remove()
    np.savez_compressed(os.path.join(out_dir, "tiny_sam_plain.npz"), mask_logits=masks.astype(np.float32),
                        tap_hs=taps["hs"].astype(np.float32), tap_low_res_masks=taps["low_res_masks"].astype(np.float32))
    print("sam_plain: masks", masks.shape, "std %.3f min %.3f max %.3f" % (masks.std(), masks.min(), masks.max()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--demo-digest", action="store_true")
    ap.add_argument("--only-sam-plain", action="store_true")
    ap.add_argument("--only-outliers", action="store_true")
    ap.add_argument("--only-hires-digest", action="store_true")
    ap.add_argument("--only-n3", action="store_true")
    ap.add_argument("--skip-tiny", action="store_true")
    ap.add_argument("--only-demo-digest", action="store_true")
    ap.add_argument("--only-demo-alpha-digest", action="store_true")
    ap.add_argument("--only-demo-seed1-digest", action="store_true", help="demo geometry, weight seed 1 (default 4 images) -> demo_digest_seed1.npz")
    ap.add_argument("--only-demo-outlier-digest", action="store_true", help="demo geometry, synth.apply_outliers weights (default 2 images) -> demo_digest_outliers.npz")
    ap.add_argument("--only-tiny", action="store_true")
    ap.add_argument("--only-train-branch", action="store_true")
    ap.add_argument("--images", type=int, default=None,
                    help="digest runs: number of images (default 16 for the demo digest, 4 for the 1536^2 digest)")
    ap.add_argument("--check", action="store_true",
                    help="digest runs: do not write; compare the reference's fresh output with the first --images images "
                         "of the committed file, bit for bit")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    mods = install_reference()
    if args.only_sam_plain:
        sam_plain(args.out, mods)
        sys.exit(0)
    if args.only_outliers:
        tiny_outliers(args.out, mods)
        sys.exit(0)
    if args.only_n3:
        n3_openai(args.out, mods)
        sys.exit(0)
    if args.only_hires_digest:
        hires_digest(args.out, mods, args.images or 4, args.check)
        sys.exit(0)
    if args.only_demo_digest:
        demo_digest(args.out, mods, args.images or 16, args.check)
        sys.exit(0)
    if args.only_demo_seed1_digest:
        demo_digest(args.out, mods, args.images or 4, args.check, seed=1, name="demo_digest_seed1.npz")
        sys.exit(0)
    if args.only_demo_outlier_digest:
        demo_digest(args.out, mods, args.images or 2, args.check, outliers=True, name="demo_digest_outliers.npz")
        sys.exit(0)
    if args.only_demo_alpha_digest:
        demo_alpha_digest(args.out, mods, args.images or 1, args.check)
        sys.exit(0)
    if args.only_tiny:
        tiny(args.out, mods)
        sys.exit(0)
    if args.only_train_branch:
        train_branch(args.out, mods, args.check)
        sys.exit(0)
    tokens(args.out, mods)
    if not args.skip_tiny:
        tiny(args.out, mods)
    sam_plain(args.out, mods)
    if args.demo_digest:
        demo_digest(args.out, mods, args.images or 16, args.check)
