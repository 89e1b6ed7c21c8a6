"""Parity gates against the digests of the REFERENCE's own outputs (tests/golden/demo_digest.npz, hires1536_digest.npz;
written by tools/make_golden.py from /root/reference, one B = 1 forward per image): shared by bench.py and tests/.

Gate (BASELINE.json north_star): mask logits and class logits within 1e-3 abs of the reference's fp32 CPU forward on the
sampled positions (4096 common ones; from round 6 on also 65536 more and each image's 4096 pixels nearest the decision boundary), mask IoU >= 0.999 on the full-resolution sign bits, identical predictions -- for EVERY image handed in.
Checker code only: nothing here runs on the product path."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

TOL = 1e-3
IOU_MIN = 0.999


def load(path: str) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def golden_path(name: str) -> str:
    from . import REPO_DIR
    return os.path.join(REPO_DIR, "tests", "golden", name)


def n_images(dg: Dict[str, np.ndarray]) -> int:
    return int(dg["pred"].shape[0]) if "pred" in dg else int(dg["samples"].shape[0])


def check_cascade(masks, pred, logits, dg: Dict[str, np.ndarray], image_ids: Sequence[int],
                  pass1_logits=None) -> dict:
    """masks (B,1,S,S) mask logits, pred (B,), logits (B,n_cls) of images `image_ids` (indices into the digest; ids the
    digest does not hold are skipped).  Returns the worst figures over the checked images and `ok`; when NO image could be
    checked `ok` is None (neither pass nor fail: a caller that tests `r["ok"]` alone must not read that as verified)."""
    m = masks.detach().float().cpu().numpy()
    lg = logits.detach().float().cpu().numpy()
    pr = pred.detach().cpu().numpy()
    p1 = pass1_logits.detach().float().cpu().numpy() if pass1_logits is not None else None
    n = n_images(dg)
    per, checked = [], []
    for b, iid in enumerate(image_ids):
        if iid >= n:
            continue
        mb = m[b].reshape(-1)
        ref_bits = np.unpackbits(dg["mask_bits"][iid])[:mb.size].astype(bool)
        got_bits = mb > 0
        inter, union = float((got_bits & ref_bits).sum()), float((got_bits | ref_bits).sum())
        # mask logits: the 4096 common positions of rounds 1-5, and (digests written from round 6 on) 65536 more common positions and the
        # image's own 4096 positions of smallest |logit| -- where the arithmetic could flip a sign
        errs = {"sparse": float(np.abs(mb[dg["sample_idx"]] - dg["mask_samples"][iid]).max())}
        if "dense_idx" in dg:
            errs["dense"] = float(np.abs(mb[dg["dense_idx"]] - dg["dense_samples"][iid]).max())
            errs["near"] = float(np.abs(mb[dg["near_idx"][iid]] - dg["near_samples"][iid]).max())
        rec = {"image": int(iid), "iou": inter / max(union, 1.0), "mask_err": max(errs.values()), "mask_err_sets": errs,
               "logit_err": float(np.abs(lg[b] - dg["class_logits"][iid]).max()),
               "pred_equal": int(pr[b]) == int(dg["pred"][iid])}
        if p1 is not None:
            rec["pass1_logit_err"] = float(np.abs(p1[b] - dg["pass1_logits"][iid]).max())
        per.append(rec)
        checked.append(int(iid))
    if not per:
        return {"checked_images": [], "ok": None}
    out = {"checked_images": checked, "min_iou": min(r["iou"] for r in per),
           "max_abs_mask_err": max(r["mask_err"] for r in per), "max_abs_class_logit_err": max(r["logit_err"] for r in per),
           "pred_equal": all(r["pred_equal"] for r in per), "tolerance": TOL, "iou_min": IOU_MIN}
    out["mask_positions_per_image"] = int(dg["sample_idx"].size + (dg["dense_idx"].size + dg["near_idx"].shape[1] if "dense_idx" in dg else 0))
    out["max_abs_mask_err_by_set"] = {k: max(r["mask_err_sets"][k] for r in per) for k in per[0]["mask_err_sets"]}
    if p1 is not None:
        out["max_abs_pass1_logit_err"] = max(r["pass1_logit_err"] for r in per)
    out["ok"] = bool(out["min_iou"] >= IOU_MIN and out["max_abs_mask_err"] <= TOL and out["max_abs_class_logit_err"] <= TOL and
                     out["pred_equal"] and out.get("max_abs_pass1_logit_err", 0.0) <= TOL)
    return out


def check_features(feats, grid: int, dg: Dict[str, np.ndarray], image_ids: Sequence[int], samples_key: str, idx_key: str,
                   cmean_key: str) -> dict:
    """Encoder output, token-major f32 [B*grid*grid][C] (the engine's NHWC layout), against the reference's (1,C,grid,grid)
    samples and per-channel means of images `image_ids`."""
    f = feats.detach().float().cpu().numpy()
    C = f.shape[-1]
    f = f.reshape(-1, grid, grid, C)
    n = int(dg[samples_key].shape[0])
    errs, cms, checked = [], [], []
    for b, iid in enumerate(image_ids):
        if iid >= n:
            continue
        nchw = np.ascontiguousarray(f[b].transpose(2, 0, 1))
        errs.append(float(np.abs(nchw.reshape(-1)[dg[idx_key]] - dg[samples_key][iid]).max()))
        cms.append(float(np.abs(nchw.mean(axis=(1, 2)) - dg[cmean_key][iid]).max()))
        checked.append(int(iid))
    if not errs:
        return {"checked_images": [], "ok": None}
    return {"checked_images": checked, "max_abs_feature_err": max(errs), "max_abs_channel_mean_err": max(cms), "tolerance": TOL,
            "ok": bool(max(errs) <= TOL and max(cms) <= TOL)}


def check_demo_features(feats, grid: int, dg, image_ids: Sequence[int]) -> dict:
    return check_features(feats, grid, dg, image_ids, "feat_samples", "feat_idx", "feat_channel_mean")


def check_hires_features(feats, grid: int, dg, image_ids: Sequence[int]) -> dict:
    return check_features(feats, grid, dg, image_ids, "samples", "sample_idx", "channel_mean")


def rank_batches(rank: int, batch: int, n_digest: int, n_batches: int = 2):
    """Image ids of the batches rank `rank` alternates in bench.py's timed loop: the digest's images rotated by the rank, batch k =
    {(rank + k * batch + i) mod n_digest}.  Every id is a digest image, so every rank's parity check is a check against the reference
    (SURVEY.md section 4(iv)); ranks differ in order and, for batch < n_digest, in content."""
    return [[(rank + k * batch + i) % n_digest for i in range(batch)] for k in range(n_batches)]
