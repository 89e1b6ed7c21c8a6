#!/usr/bin/env python3
"""Golden vectors for the evaluation tail, made by running the REFERENCE's own classes on CPU
(build container only; nothing of the reference is copied):

  * recorder/ovcos_metricer.py `IOU` (:126-180) -- adaptive and changeable IoU of seeded uint8 masks.  The module
    subclasses pysodmetrics 1.4.2, which is not installed here; `py_sod_metrics` is stubbed with empty base classes
    and with `prepare_data` / `get_adaptive_threshold` taken from oracle/metrics_oracle.py, so the IoU arithmetic
    that runs is the reference's, the normalisation in front of it is the restatement (stated in the oracle header).
  * recorder/new_evaluator.py `Classification` (:23-122) -- top-1 / top-5 / macro-F1 on seeded scores.

Usage:  python tools/make_evaltail_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CVLM_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402

from oracle import metrics_oracle as mo  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seeded_case(seed: int, h: int, w: int, kind: str):
    """uint8 prediction / ground truth pairs shaped like camouflage masks: a soft blob over a binary blob."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    cy, cx, r = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w, rng.uniform(0.1, 0.3) * min(h, w)
    d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
    gt = np.where(d < r, 255, 0).astype(np.uint8)
    soft = 1 / (1 + np.exp((d - r * rng.uniform(0.8, 1.2)) / rng.uniform(1.5, 6.0))) + rng.normal(0, 0.05, (h, w))
    pre = (np.clip(soft, 0, 1) * 255).astype(np.uint8)
    if kind == "empty_gt":
        gt[:] = 0
    elif kind == "full_gt":
        gt[:] = 255
    elif kind == "flat_pred":
        pre[:] = 77
    elif kind == "narrow":
        pre = (pre // 4 + 60).astype(np.uint8)
    return pre, gt


CASES = [(1, 97, 131, "blob"), (2, 64, 64, "blob"), (3, 120, 75, "narrow"), (4, 50, 70, "empty_gt"),
         (5, 40, 40, "full_gt"), (6, 33, 47, "flat_pred"), (7, 256, 192, "blob")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    base = type("Base", (), {})
    _stub("py_sod_metrics")
    _stub("py_sod_metrics.sod_metrics", MAE=base, Emeasure=base, Fmeasure=base, Smeasure=base, WeightedFmeasure=base)
    _stub("py_sod_metrics.utils", TYPE=np.float64, get_adaptive_threshold=mo.adaptive_threshold, prepare_data=mo.prepare_data)
    _stub("utils", log=lambda *a, **k: None)
    metricer = _load("ref_ovcos_metricer", "recorder/ovcos_metricer.py")
    evaluator = _load("ref_new_evaluator", "recorder/new_evaluator.py")

    out = {}
    for i, (seed, h, w, kind) in enumerate(CASES):
        pre, gt = seeded_case(seed, h, w, kind)
        out[f"case{i}_pre"], out[f"case{i}_gt"] = pre, gt
        for same in (True, False):
            iou = metricer.IOU()
            iou.step(pre, gt, "a", "a" if same else "b")
            res = iou.get_results()["iou"]
            tag = f"case{i}_{'same' if same else 'diff'}"
            out[f"{tag}_adp"] = np.asarray(res["adp"], dtype=np.float64).reshape(-1)
            out[f"{tag}_curve"] = np.asarray(res["curve"], dtype=np.float64).reshape(-1)
    out["cases"] = np.asarray([[s, h, w] for s, h, w, _ in CASES], dtype=np.int64)
    out["kinds"] = np.asarray([k for *_, k in CASES])

    rng = np.random.default_rng(11)
    cls = evaluator.Classification()
    cls.reset()
    scores_all, labels_all = [], []
    for b in (1, 4, 7, 16):
        scores = rng.normal(size=(b, 14)).astype(np.float32)
        labels = rng.integers(0, 14, size=b)
        labels[::3] = scores[::3].argmax(axis=1)          # make a share of the rows correct
        cls.process(torch.from_numpy(scores), torch.from_numpy(labels))
        scores_all.append(scores)
        labels_all.append(labels)
    res = cls.evaluate()
    out["cls_scores"] = np.concatenate(scores_all)
    out["cls_labels"] = np.concatenate(labels_all).astype(np.int64)
    out["cls_batches"] = np.asarray([1, 4, 7, 16], dtype=np.int64)
    out["cls_result"] = np.asarray([res["accuracy"], res["error_rate"], res["top5"], res["macro_f1"]], dtype=np.float64)
    out["cls_counts"] = np.asarray([cls._correct, cls._correct_5, cls._total], dtype=np.int64)
    os.makedirs(args.out, exist_ok=True)
    path = os.path.join(args.out, "evaltail.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("cls")}, res)


if __name__ == "__main__":
    main()
