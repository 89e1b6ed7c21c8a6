#!/usr/bin/env python3
"""Golden vectors for the evaluation tail, made by running the REFERENCE's own classes on CPU
(build container only; nothing of the reference is copied):

  * recorder/ovcos_metricer.py `OVCOSMetricer.step / show / _get_raw_results` (:257-307) with its six metric classes
    (:8-180).  They subclass `py_sod_metrics.sod_metrics.*` -- pysodmetrics 1.4.2 is not installed, but the reference
    ships the same classes in-tree as recorder/sod_metric.py:39-581 (live code: utils.py:143-165 `calc_cod`, called at
    test_ovcos_maskdecoder_edge.py:105).  `py_sod_metrics.sod_metrics` / `py_sod_metrics.utils` are therefore registered
    as modules whose members ARE those in-tree classes and functions (`prepare_data = _prepare_data`,
    `get_adaptive_threshold = _get_adaptive_threshold`, `TYPE = np.float64`): every line of arithmetic that runs is the
    reference's.  `cv2` (imported, never called on this path) and `tensorboardX` are empty stubs.
  * utils.py `calc_cod` (:143-165) -- the Sm / Em / wFm / MAE of the float probability map the loop also logs.
  * recorder/new_evaluator.py `Classification` (:23-122) -- top-1 / top-5 / macro-F1 on seeded scores.

Usage:  python tools/make_evaltail_golden.py [--out tests/golden] [--check]
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


METRIC_KEYS = ("sm", "wfm", "mae", "fm_adp", "fm_curve", "em_adp", "em_curve", "iou_adp", "iou_curve")
SHOW_KEYS = ("sm", "wfm", "mae", "adpfm", "maxfm", "avgfm", "adpem", "maxem", "avgem", "adpiou", "maxiou", "avgiou")
# the order `OVCOSMetricer.step` is driven in for the aggregated `show()` vector: (case index, predicted class == gt class)
SHOW_SEQUENCE = [(0, True), (1, True), (2, False), (3, True), (4, True), (5, True), (6, True), (0, False), (6, True)]


def load_reference():
    """-> (recorder.sod_metric, recorder.ovcos_metricer, recorder.new_evaluator, utils) of the reference, see the header."""
    _stub("cv2")
    _stub("tensorboardX", SummaryWriter=object)
    pkg = _stub("recorder")                              # the package's own __init__ pulls matplotlib / torchvision
    pkg.__path__ = [os.path.join(REF, "recorder")]
    sod = _load("recorder.sod_metric", "recorder/sod_metric.py")
    sys.modules["recorder.sod_metric"] = sod
    pkg.sod_metric = sod
    _stub("py_sod_metrics")
    _stub("py_sod_metrics.sod_metrics", MAE=sod.MAE, Emeasure=sod.Emeasure, Fmeasure=sod.Fmeasure, Smeasure=sod.Smeasure,
          WeightedFmeasure=sod.WeightedFmeasure)
    _stub("py_sod_metrics.utils", TYPE=np.float64, get_adaptive_threshold=sod._get_adaptive_threshold, prepare_data=sod._prepare_data)
    utils = _load("utils", "utils.py")                   # the real module: `calc_cod`, and `log` for new_evaluator
    sys.modules["utils"] = utils
    utils.log = lambda *a, **k: None
    metricer = _load("recorder.ovcos_metricer", "recorder/ovcos_metricer.py")
    evaluator = _load("recorder.new_evaluator", "recorder/new_evaluator.py")
    return sod, metricer, evaluator, utils


def per_image(metricer_mod, pre, gt, same):
    """one `OVCOSMetricer.step` (ovcos_metricer.py:269-272) -> the nine per-image values its metric objects hold"""
    m = metricer_mod.OVCOSMetricer(class_names=["a", "b"])
    m.step(pre=pre, gt=gt, pre_cls="a", gt_cls="a" if same else "b", gt_path="golden")
    r = {n: o.get_results()[n] for n, o in m.metric_objs.items()}
    f = lambda a: np.asarray(a, dtype=np.float64).reshape(-1)  # noqa: E731
    return {"sm": f(r["sm"]), "wfm": f(r["wfm"]), "mae": f(r["mae"]), "fm_adp": f(r["fm"]["adp"]), "fm_curve": f(r["fm"]["curve"]),
            "em_adp": f(r["em"]["adp"]), "em_curve": f(r["em"]["curve"]), "iou_adp": f(r["iou"]["adp"]), "iou_curve": f(r["iou"]["curve"])}


def cod_batch():
    """what `calc_cod` sees (test_ovcos_maskdecoder_edge.py:103-105): sigmoid probabilities and a {0, 1} float ground truth,
    (B,1,H,W) float32; image 2 has an empty ground truth, image 3 a full one"""
    rng = np.random.default_rng(21)
    b, h, w = 4, 96, 128
    yy, xx = np.mgrid[0:h, 0:w]
    pred = np.empty((b, 1, h, w), dtype=np.float32)
    gt = np.zeros((b, 1, h, w), dtype=np.float32)
    for k in range(b):
        cy, cx, r = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w, rng.uniform(0.15, 0.3) * h
        d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
        logit = (r - d) / rng.uniform(1.0, 5.0) + rng.normal(0, 0.7, (h, w))
        pred[k, 0] = (1 / (1 + np.exp(-logit))).astype(np.float32)
        gt[k, 0] = (d < r * rng.uniform(0.8, 1.1)).astype(np.float32)
    gt[2] = 0
    gt[3] = 1
    return pred, gt


def build():
    sod, metricer, evaluator, utils = load_reference()
    out = {}
    for i, (seed, h, w, kind) in enumerate(CASES):
        pre, gt = seeded_case(seed, h, w, kind)
        out[f"case{i}_pre"], out[f"case{i}_gt"] = pre, gt
        for same in (True, False):
            tag = f"case{i}_{'same' if same else 'diff'}"
            for k, v in per_image(metricer, pre, gt, same).items():
                out[f"{tag}_{k}"] = v
    out["cases"] = np.asarray([[s, h, w] for s, h, w, _ in CASES], dtype=np.int64)
    out["kinds"] = np.asarray([k for *_, k in CASES])

    # the aggregate: ONE metricer fed a sequence of images, `show()` as the loop prints it (test_ovcos_maskdecoder_edge.py:141)
    m = metricer.OVCOSMetricer(class_names=["a", "b"], metric_names=("sm", "wfm", "mae", "fm", "em", "iou"))
    for i, same in SHOW_SEQUENCE:
        m.step(pre=out[f"case{i}_pre"], gt=out[f"case{i}_gt"], pre_cls="a", gt_cls="a" if same else "b")
    raw, shown = m.get_step_results(), m.show()
    out["show_sequence"] = np.asarray([[i, int(s)] for i, s in SHOW_SEQUENCE], dtype=np.int64)
    out["show_raw"] = np.asarray([float(raw[k]) for k in SHOW_KEYS], dtype=np.float64)
    out["show_rounded"] = np.asarray([float(shown[k]) for k in SHOW_KEYS], dtype=np.float64)

    pred, gt = cod_batch()
    sm, em, wfm, mae = utils.calc_cod(torch.from_numpy(pred), torch.from_numpy(gt))
    out["cod_pred"], out["cod_gt"] = pred, gt.astype(np.uint8)
    out["cod_result"] = np.asarray([sm, em, wfm, mae], dtype=np.float64)
    out["cod_per_image"] = np.asarray([[float(v) for v in utils.calc_cod(torch.from_numpy(pred[k:k + 1]), torch.from_numpy(gt[k:k + 1]))]
                                       for k in range(len(pred))], dtype=np.float64)

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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--check", action="store_true", help="regenerate and compare with the committed file instead of writing")
    args = ap.parse_args()
    out = build()
    path = os.path.join(args.out, "evaltail.npz")
    if args.check:
        with np.load(path) as z:
            assert sorted(z.files) == sorted(out), sorted(set(z.files) ^ set(out))
            bad = [k for k in z.files if not np.array_equal(z[k], out[k])]
        print("IDENTICAL" if not bad else f"DIFFERENT: {bad}", f"({len(out)} keys)")
        sys.exit(1 if bad else 0)
    os.makedirs(args.out, exist_ok=True)
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "keys; show():", dict(zip(SHOW_KEYS, out["show_rounded"].tolist())), "calc_cod:", out["cod_result"].tolist())


if __name__ == "__main__":
    main()
