"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the evaluation tail.  Not imported by the product.

Per-pixel numpy restatement of what the reference does after `infer_test` for one image:

  * `mask_to_u8`      test_ovcos_maskdecoder_edge.py:103,116-130 -- sigmoid, cv2.resize(INTER_LINEAR) on float32,
                      `(pred * 255).astype(np.uint8)`.
  * `ovcos_metrics`   recorder/ovcos_metricer.py:8-180 `*.step` -- thin subclasses of **pysodmetrics 1.4.2**
                      (requirements.txt:23; the package itself is NOT in /root/reference and not installed here).
  * `iou_*`           recorder/ovcos_metricer.py:126-180 -- the reference's own IOU class.
  * `Classification`  recorder/new_evaluator.py:47-59,68-71 -- top-1 / top-5.

Pinning (round 4): **every metric is pinned to the reference's own code run in this container**
(tests/golden/evaltail.npz, tools/make_evaltail_golden.py): the reference carries the pysodmetrics classes in-tree as
recorder/sod_metric.py:39-581 (the code `utils.calc_cod` runs, utils.py:143-165); the generator registers them under the
`py_sod_metrics` names recorder/ovcos_metricer.py imports and drives the real `OVCOSMetricer.step / show`, `calc_cod` and
`Classification`.  Gated per image (Sm, wFm, MAE, adaptive and 256-point F / E / IoU curves, class match and mismatch,
empty / full ground truth, flat prediction, non-square sizes) and on the aggregated `show()` dict.  Only `cv2.resize`
(OpenCV 4.8's float32 linear path, restated in `resize_linear_f32`; the package is not installed and nothing of it is in
/root/reference) stays **parity unpinned**.  The weighted F-measure runs on scipy's exact distance transform and
convolution, the same routines the reference calls (sod_metric.py:4-5).
"""
from __future__ import annotations

import numpy as np

_EPS = np.spacing(1)


# ---- mask -> uint8 ---------------------------------------------------------------------------------------------------
def _axis_table(n_src: int, n_dst: int, clamp_weight: bool):
    scale = float(n_src) / float(n_dst)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    i0 = np.floor(f).astype(np.int64)
    frac = (f - i0.astype(np.float32)).astype(np.float32)
    if clamp_weight:                       # columns: cv2 zeroes the weight where the window leaves the row
        lo, hi = i0 < 0, i0 >= n_src - 1
        frac = np.where(lo | hi, np.float32(0), frac)
        i0 = np.where(lo, 0, np.where(hi, n_src - 1, i0))
        i1 = np.minimum(i0 + 1, n_src - 1)
    else:                                  # rows: indices clamped, weight kept
        i1 = np.clip(i0 + 1, 0, n_src - 1)
        i0 = np.clip(i0, 0, n_src - 1)
    return i0, i1, (np.float32(1) - frac).astype(np.float32), frac.astype(np.float32)


def resize_linear_f32(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=INTER_LINEAR) for a 2-D float32 array; identity when sizes match
    (test_ovcos_maskdecoder_edge.py:37-43)."""
    img = np.asarray(img, dtype=np.float32)
    if img.shape == (h, w):
        return img
    x0, x1, a0, a1 = _axis_table(img.shape[1], w, True)
    y0, y1, b0, b1 = _axis_table(img.shape[0], h, False)
    rows = img[:, x0] * a0[None, :] + img[:, x1] * a1[None, :]
    return (rows[y0] * b0[:, None] + rows[y1] * b1[:, None]).astype(np.float32)


def sigmoid_f32(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def mask_to_u8(logits: np.ndarray, h: int, w: int) -> np.ndarray:
    return (resize_linear_f32(sigmoid_f32(logits), h, w) * 255).astype(np.uint8)


# ---- recorder/sod_metric.py:12-581 (= pysodmetrics 1.4.2) ----------------------------------------------------------------------------------------------
def prepare_data(pred: np.ndarray, gt: np.ndarray):
    gt = gt > 128
    pred = pred / 255
    if pred.max() != pred.min():
        pred = (pred - pred.min()) / (pred.max() - pred.min())
    return pred, gt


def adaptive_threshold(m: np.ndarray, max_value: float = 1) -> float:
    return min(2 * m.mean(), max_value)


def mae(pred, gt) -> float:
    return float(np.mean(np.abs(pred - gt)))


def fm_adaptive(pred, gt, beta=0.3) -> float:
    b = pred >= adaptive_threshold(pred)
    inter = b[gt].sum()
    if inter == 0:
        return 0.0
    pre = inter / np.count_nonzero(b)
    rec = inter / np.count_nonzero(gt)
    return float((1 + beta) * pre * rec / (beta * pre + rec))


def _cum_hists(pred, gt):
    q = (pred * 255).astype(np.uint8)
    bins = np.linspace(0, 256, 257)
    fg, _ = np.histogram(q[gt], bins=bins)
    bg, _ = np.histogram(q[~gt], bins=bins)
    return np.cumsum(np.flip(fg)), np.cumsum(np.flip(bg))


def fm_changeable(pred, gt, beta=0.3) -> np.ndarray:
    tp, fp = _cum_hists(pred, gt)
    ps = tp + fp
    ps[ps == 0] = 1
    t = max(np.count_nonzero(gt), 1)
    precisions, recalls = tp / ps, tp / t
    num = (1 + beta) * precisions * recalls
    den = np.where(num == 0, 1, beta * precisions + recalls)
    return num / den


def _em_from_counts(fg_fg, fg_bg, gt_fg, size):
    """fg_fg / fg_bg: predicted-foreground counts inside gt foreground / background (scalars or arrays)."""
    pred_fg = fg_fg + fg_bg
    pred_bg = size - pred_fg
    if gt_fg == 0:
        s = pred_bg
    elif gt_fg == size:
        s = pred_fg
    else:
        bg_fg = gt_fg - fg_fg
        bg_bg = pred_bg - bg_fg
        mp, mg = pred_fg / size, gt_fg / size
        s = 0
        for part, (dp, dg) in zip((fg_fg, fg_bg, bg_fg, bg_bg), ((1 - mp, 1 - mg), (1 - mp, 0 - mg), (0 - mp, 1 - mg), (0 - mp, 0 - mg))):
            align = 2 * (dp * dg) / (dp ** 2 + dg ** 2 + _EPS)
            s = s + (align + 1) ** 2 / 4 * part
    return s / (size - 1 + _EPS)


def em_adaptive(pred, gt) -> float:
    b = pred >= adaptive_threshold(pred)
    return float(_em_from_counts(np.count_nonzero(b & gt), np.count_nonzero(b & ~gt), np.count_nonzero(gt), gt.size))


def em_changeable(pred, gt) -> np.ndarray:
    tp, fp = _cum_hists(pred, gt)
    return np.asarray(_em_from_counts(tp, fp, np.count_nonzero(gt), gt.size), dtype=np.float64)


def _s_object(p, g) -> float:
    x = np.mean(p[g == 1])
    sigma = np.std(p[g == 1], ddof=1)
    return 2 * x / (np.power(x, 2) + 1 + sigma + _EPS)


def _ssim(p, g) -> float:
    n = p.size
    x, y = np.mean(p), np.mean(g)
    sx = np.sum((p - x) ** 2) / (n - 1)
    sy = np.sum((g - y) ** 2) / (n - 1)
    sxy = np.sum((p - x) * (g - y)) / (n - 1)
    alpha = 4 * x * y * sxy
    beta = (x ** 2 + y ** 2) * (sx + sy)
    if alpha != 0:
        return alpha / (beta + _EPS)
    return 1 if beta == 0 else 0


def centroid(gt: np.ndarray):
    h, w = gt.shape
    if np.count_nonzero(gt) == 0:
        x, y = np.round(w / 2), np.round(h / 2)
    else:
        y, x = np.argwhere(gt).mean(axis=0).round()
    return int(x) + 1, int(y) + 1


def sm(pred, gt, alpha=0.5) -> float:
    with np.errstate(all="ignore"):
        y = np.mean(gt)
        if y == 0:
            return float(1 - np.mean(pred))
        if y == 1:
            return float(np.mean(pred))
        g = gt                                               # bool, as in sod_metric.py:213-221: float32 maps stay float32
        obj = y * _s_object(pred * g, g) + (1 - y) * _s_object((1 - pred) * (1 - g), 1 - g)
        cx, cy = centroid(gt)
        h, w = gt.shape
        area = h * w
        w1, w2, w3 = cx * cy / area, cy * (w - cx) / area, (h - cy) * cx / area
        w4 = 1 - w1 - w2 - w3
        parts = ((slice(0, cy), slice(0, cx)), (slice(0, cy), slice(cx, w)), (slice(cy, h), slice(0, cx)), (slice(cy, h), slice(cx, w)))
        reg = sum(wk * _ssim(pred[s], g[s]) for wk, s in zip((w1, w2, w3, w4), parts))
        return float(max(0, alpha * obj + (1 - alpha) * reg))


def gauss2d(shape=(7, 7), sigma=5.0) -> np.ndarray:
    """pysodmetrics `matlab_style_gauss2D` (MATLAB fspecial('gaussian', 7, 5))."""
    m, n = [(ss - 1) / 2 for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    sumh = h.sum()
    if sumh != 0:
        h /= sumh
    return h


def wfm(pred, gt, beta=1.0) -> float:
    """`WeightedFmeasure.cal_wfm` (recorder/sod_metric.py:516-560, Margolin et al.), as called by ovcos_metricer.py:49-66
    with beta = 1 and by `calc_cod` with the class default 0.3 (sod_metric.py:491); scipy's exact Euclidean distance
    transform and `convolve`, as there."""
    from scipy.ndimage import convolve, distance_transform_edt as bwdist
    if np.all(~gt):
        return 0.0
    dst, idxt = bwdist(gt == 0, return_indices=True)
    e = np.abs(pred - gt)
    et = np.copy(e)
    et[gt == 0] = et[idxt[0][gt == 0], idxt[1][gt == 0]]
    ea = convolve(et, weights=gauss2d((7, 7), sigma=5), mode="constant", cval=0)
    min_e_ea = np.where(gt & (ea < e), ea, e)
    b = np.where(gt == 0, 2 - np.exp(np.log(0.5) / 5 * dst), np.ones_like(gt, dtype=np.float64))
    ew = min_e_ea * b
    tpw = np.sum(gt) - np.sum(ew[gt == 1])
    fpw = np.sum(ew[gt == 0])
    r = 1 - np.mean(ew[gt == 1])
    p = tpw / (tpw + fpw + _EPS)
    return float((1 + beta) * r * p / (r + beta * p + _EPS))


# ---- recorder/ovcos_metricer.py:126-180 (the reference's own IOU class) ---------------------------------------------
def iou_adaptive(pred, gt) -> float:
    b = pred >= adaptive_threshold(pred)
    union = np.count_nonzero(b | gt)
    return 0.0 if union == 0 else float(np.count_nonzero(b & gt) / union)


def iou_changeable(pred, gt) -> np.ndarray:
    tp, fp = _cum_hists(pred, gt)
    fn = np.count_nonzero(gt) - tp
    den = np.array(tp + fn + fp, dtype=np.float64)
    np.divide(tp, den, out=den, where=den != 0)
    return den


def ovcos_metrics(pre_u8: np.ndarray, gt_u8: np.ndarray, same_class: bool = True) -> dict:
    """One `OVCOSMetricer.step` (ovcos_metricer.py:269-272) without wfm: per-image values, zeroed (MAE: 1) when the
    predicted class differs (`:18-19`, `:36-37`, `:83-85`, `:109-111`, `:139-141`)."""
    assert pre_u8.dtype == np.uint8 and gt_u8.dtype == np.uint8 and pre_u8.shape == gt_u8.shape
    pred, gt = prepare_data(pre_u8, gt_u8)
    out = {"sm": sm(pred, gt), "wfm": wfm(pred, gt), "mae": mae(pred, gt), "fm_adp": fm_adaptive(pred, gt), "fm_curve": fm_changeable(pred, gt),
           "em_adp": em_adaptive(pred, gt), "em_curve": em_changeable(pred, gt), "iou_adp": iou_adaptive(pred, gt),
           "iou_curve": iou_changeable(pred, gt)}
    if not same_class:
        out = {k: (np.ones_like(v) if k == "mae" else np.zeros_like(v)) * 1.0 for k, v in out.items()}
    return out


def aggregate(steps: list) -> dict:
    """`OVCOSMetricer._get_raw_results` (ovcos_metricer.py:277-297) over a list of `ovcos_metrics` results."""
    res = {"sm": float(np.mean([s["sm"] for s in steps])), "wfm": float(np.mean([s["wfm"] for s in steps])),
           "mae": float(np.mean([s["mae"] for s in steps]))}
    for m in ("fm", "em", "iou"):
        curve = np.stack([np.asarray(s[f"{m}_curve"], dtype=np.float64) for s in steps]).mean(axis=0)
        res[f"adp{m}"] = float(np.mean([s[f"{m}_adp"] for s in steps]))
        res[f"max{m}"] = float(curve.max())
        res[f"avg{m}"] = float(curve.mean())
    return res


def calc_cod(y_pred: np.ndarray, y_true: np.ndarray):
    """utils.py:143-165 `calc_cod`: (B,1,H,W) float32 probabilities and {0, 1} ground truth -> (sm, em, wfm, mae), each the
    batch mean of the in-tree sod_metric classes fed `y * 255` as float32 (no uint8 step: Sm / MAE / wFm see the continuous
    map, only the E curve quantises, sod_metric.py:420); wFm at the class default beta = 0.3; em = mean of the mean curve."""
    sms, ems, wfms, maes = [], [], [], []
    for p, t in zip(np.asarray(y_pred)[:, 0], np.asarray(y_true)[:, 0]):
        pred, gt = prepare_data(p * 255, t * 255)
        sms.append(sm(pred, gt))
        ems.append(em_changeable(pred, gt))
        wfms.append(wfm(pred, gt, beta=0.3))
        maes.append(mae(pred, gt))
    return (float(np.mean(np.array(sms, np.float64))), float(np.mean(np.array(ems, np.float64), axis=0).mean()),
            float(np.mean(np.array(wfms, np.float64))), float(np.mean(np.array(maes, np.float64))))


# ---- recorder/new_evaluator.py:47-59,68-71 --------------------------------------------------------------------------
def classification(scores: np.ndarray, labels: np.ndarray):
    """-> (pred [B], top-1 hits, top-5 hits); ties resolved towards the lower class index."""
    scores = np.asarray(scores, dtype=np.float32)
    pred = scores.argmax(axis=1)
    order = np.argsort(-scores, axis=1, kind="stable")[:, :5]
    top5 = int(sum(int(l) in row.tolist() for l, row in zip(labels, order)))
    return pred, int((pred == labels).sum()), top5
