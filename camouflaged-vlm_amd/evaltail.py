"""Evaluation tail on the device: what the reference does with `infer_test`'s output.

The reference pulls every 1024x1024 float mask to the host, resizes it with cv2 and walks it several times in
numpy (test_ovcos_maskdecoder_edge.py:116-136, recorder/ovcos_metricer.py).  Every metric it then computes except the
weighted F-measure only depends on the uint8 mask through counts: per S-measure quadrant, per ground-truth class, per
level.  `mask_counts` produces exactly those counters on the GPU (`csrc/evaltail.hip`); the functions below turn the
8 KB of counters per image into MAE, adaptive / changeable F, E and IoU measures and the S-measure in float64.
The weighted F-measure is spatial (distance transform, 7x7 Gaussian): `mask_wfm_sums` runs it on the GPU in float64
and returns three sums per image.

There is no CPU path: `mask_counts` and `DeviceClassification` raise if the HIP library is missing.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import hip

_EPS = np.spacing(1)
CURVE_METRICS = ("fm", "em", "iou")
SUPPORTED = ("sm", "wfm", "mae", "fm", "em", "iou")


# ---- device side -----------------------------------------------------------------------------------------------------
def mask_to_u8(logits: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """logits (N,1,Hs,Ws) or (N,Hs,Ws) f32 on the GPU -> uint8 (N,h,w): sigmoid, bilinear resize, *255, truncate."""
    if logits.dim() == 4:
        logits = logits[:, 0]
    logits = logits.float().contiguous()
    out = torch.empty((logits.shape[0], h, w), dtype=torch.uint8, device=logits.device)
    hip.mask_to_u8(logits, h, w, out)
    return out


def mask_counts(pre: torch.Tensor, gt: torch.Tensor):
    """pre / gt uint8 (N,h,w) on the GPU -> (stats int64 (N,3), hist int32 (N,4,2,256)) device tensors."""
    n = pre.shape[0]
    stats = torch.empty((n, 3), dtype=torch.int64, device=pre.device)
    hist = torch.empty((n, 4, 2, 256), dtype=torch.int32, device=pre.device)
    hip.mask_joint_hist(pre.contiguous(), gt.contiguous(), stats, hist)
    return stats, hist


_GAUSS: Dict[str, torch.Tensor] = {}


def _gauss49(device) -> torch.Tensor:
    """fspecial('gaussian', 7, 5) as pysodmetrics builds it: exp(-(x^2+y^2)/(2 sigma^2)), tiny entries zeroed, sum 1."""
    key = str(device)
    if key not in _GAUSS:
        y, x = np.ogrid[-3:4, -3:4]
        k = np.exp(-(x * x + y * y) / (2.0 * 5.0 * 5.0))
        k[k < np.finfo(k.dtype).eps * k.max()] = 0
        k /= k.sum()
        _GAUSS[key] = torch.from_numpy(np.ascontiguousarray(k.reshape(-1))).to(device)
    return _GAUSS[key]


def mask_wfm_sums(pre: torch.Tensor, gt: torch.Tensor, hist: torch.Tensor) -> torch.Tensor:
    """pre / gt uint8 (N,h,w), hist from `mask_counts` -> float64 (N,3) = (sum Ew over gt, sum Ew over ~gt, |gt|)."""
    n, h, w = pre.shape
    ws = torch.empty(hip.mask_wfm_workspace_bytes(n, h, w), dtype=torch.uint8, device=pre.device)
    out = torch.empty((n, 3), dtype=torch.float64, device=pre.device)
    hip.mask_wfm(pre.contiguous(), gt.contiguous(), hist, _gauss49(pre.device), ws, out)
    return out


def wfm_from_sums(sums: np.ndarray, beta: float = 1.0) -> float:
    """WeightedFmeasure.cal_wfm's last lines (0 for an empty ground truth, ovcos_metricer.py:56-59)."""
    s_fg, s_bg, n1 = (float(v) for v in sums)
    if n1 == 0:
        return 0.0
    tpw, fpw = n1 - s_fg, s_bg
    r = 1 - s_fg / n1
    p = tpw / (tpw + fpw + _EPS)
    return (1 + beta) * r * p / (r + beta * p + _EPS)


# ---- counters -> metrics (host, float64, 2048 numbers per image) ------------------------------------------------------
def _levels(total: np.ndarray) -> np.ndarray:
    """`prepare_data` of pysodmetrics 1.4.2 per level: v / 255, min-max normalised over the levels present."""
    p = np.arange(256, dtype=np.float64) / 255
    present = np.nonzero(total.sum(axis=0))[0]
    lo, hi = p[present[0]], p[present[-1]]
    return (p - lo) / (hi - lo) if hi != lo else p


def _cumulative(norm: np.ndarray, total: np.ndarray):
    q = (norm * 255).astype(np.uint8).astype(np.int64)           # `(pred * 255).astype(np.uint8)` per level
    present = total.sum(axis=0) > 0
    fg = np.bincount(q[present], weights=total[1][present], minlength=256).astype(np.int64)
    bg = np.bincount(q[present], weights=total[0][present], minlength=256).astype(np.int64)
    return np.cumsum(fg[::-1]), np.cumsum(bg[::-1])


def _em(fg_fg, fg_bg, gt_fg, size):
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
        combos = ((1 - mp, 1 - mg), (1 - mp, 0 - mg), (0 - mp, 1 - mg), (0 - mp, 0 - mg))
        s = 0
        for part, (dp, dg) in zip((fg_fg, fg_bg, bg_fg, bg_bg), combos):
            s = s + ((2 * (dp * dg) / (dp ** 2 + dg ** 2 + _EPS)) + 1) ** 2 / 4 * part
    return s / (size - 1 + _EPS)


def _moments(cnt: np.ndarray, val: np.ndarray):
    """count-weighted (n, mean, sum of squared deviations) of `val`."""
    n = cnt.sum()
    with np.errstate(all="ignore"):
        mean = (cnt * val).sum() / n
        ssd = (cnt * (val - mean) ** 2).sum()
    return n, mean, ssd


def _ssim(cnt: np.ndarray, norm: np.ndarray):
    """S-measure region term on one quadrant; cnt (2,256)."""
    with np.errstate(all="ignore"):
        n, x, sxx = _moments(cnt.sum(axis=0), norm)
        y = cnt[1].sum() / n
        sx = sxx / (n - 1)
        sy = (cnt[1].sum() * (1 - y) ** 2 + cnt[0].sum() * y ** 2) / (n - 1)
        sxy = ((cnt[1] * (norm - x)).sum() * (1 - y) + (cnt[0] * (norm - x)).sum() * (0 - y)) / (n - 1)
        alpha = 4 * x * y * sxy
        beta = (x ** 2 + y ** 2) * (sx + sy)
        if alpha != 0:
            return alpha / (beta + _EPS)
        return 1 if beta == 0 else 0


def _sm(hist: np.ndarray, norm: np.ndarray, stats: np.ndarray, h: int, w: int, alpha: float = 0.5) -> float:
    total = hist.sum(axis=0)
    size = h * w
    n1 = int(total[1].sum())
    mean_all = float((total.sum(axis=0) * norm).sum() / size)
    if n1 == 0:
        return 1 - mean_all
    if n1 == size:
        return mean_all
    u = n1 / size
    with np.errstate(all="ignore"):
        def s_object(cnt, val):
            n, x, ssd = _moments(cnt, val)
            sigma = np.sqrt(ssd / (n - 1))
            return 2 * x / (x ** 2 + 1 + sigma + _EPS)
        obj = u * s_object(total[1], norm) + (1 - u) * s_object(total[0], 1 - norm)
        cnt, sx, sy = (int(v) for v in stats)
        cx, cy = int(np.round(sx / cnt)) + 1, int(np.round(sy / cnt)) + 1
        w1, w2, w3 = cx * cy / size, cy * (w - cx) / size, (h - cy) * cx / size
        w4 = 1 - w1 - w2 - w3
        reg = sum(wk * _ssim(hist[k], norm) for k, wk in enumerate((w1, w2, w3, w4)))
        return max(0, alpha * obj + (1 - alpha) * reg)


def metrics_from_counts(stats: np.ndarray, hist: np.ndarray, h: int, w: int, same_class: bool = True,
                        metric_names: Sequence[str] = ("sm", "mae", "fm", "em", "iou"),
                        wfm_sums: Optional[np.ndarray] = None) -> Dict[str, object]:
    """One image: stats (3,), hist (4,2,256) -> the per-image values OVCOSMetricer.step records
    (recorder/ovcos_metricer.py:13-141), zeroed (MAE: 1) when the predicted class is wrong."""
    hist = np.asarray(hist).astype(np.int64).reshape(4, 2, 256)
    total = hist.sum(axis=0)
    size = h * w
    assert int(total.sum()) == size, (int(total.sum()), size)
    norm = _levels(total)
    both = total.sum(axis=0)
    n1 = int(total[1].sum())
    thr = min(2 * float((both * norm).sum() / size), 1)
    binary = norm >= thr
    out: Dict[str, object] = {}
    if "sm" in metric_names:
        out["sm"] = float(_sm(hist, norm, np.asarray(stats), h, w))
    if "wfm" in metric_names:
        if wfm_sums is None:
            raise ValueError("'wfm' needs the sums of mask_wfm_sums")
        out["wfm"] = float(wfm_from_sums(wfm_sums))
    if "mae" in metric_names:
        out["mae"] = float(((total[1] * np.abs(norm - 1)).sum() + (total[0] * np.abs(norm)).sum()) / size)
    tp_a, fp_a = int(total[1][binary].sum()), int(total[0][binary].sum())
    tp, fp = _cumulative(norm, total)
    if "fm" in metric_names:
        beta = 0.3
        if tp_a == 0:
            out["fm_adp"] = 0.0
        else:
            pre, rec = tp_a / (tp_a + fp_a), tp_a / n1
            out["fm_adp"] = float((1 + beta) * pre * rec / (beta * pre + rec))
        ps = tp + fp
        ps[ps == 0] = 1
        precisions, recalls = tp / ps, tp / max(n1, 1)
        num = (1 + beta) * precisions * recalls
        out["fm_curve"] = num / np.where(num == 0, 1, beta * precisions + recalls)
    if "em" in metric_names:
        out["em_adp"] = float(_em(tp_a, fp_a, n1, size))
        out["em_curve"] = np.asarray(_em(tp, fp, n1, size), dtype=np.float64)
    if "iou" in metric_names:
        union = tp_a + fp_a + (n1 - tp_a)
        out["iou_adp"] = 0.0 if union == 0 else float(tp_a / union)
        den = np.array(tp + (n1 - tp) + fp, dtype=np.float64)
        np.divide(tp, den, out=den, where=den != 0)
        out["iou_curve"] = den
    if not same_class:
        out = {k: (np.ones_like(v) if k == "mae" else np.zeros_like(v)) * 1.0 for k, v in out.items()}
    return out


class DeviceMetricer:
    """`OVCOSMetricer` (recorder/ovcos_metricer.py:257-307) with the pixel passes on the GPU.

    `step_batch` queues the counter kernels and keeps the results on the device; nothing is copied back until
    `show()` / `get_step_results()`."""

    suppoted_metrics = sorted(SUPPORTED)

    def __init__(self, class_names: Sequence[str], metric_names: Sequence[str] = ("sm", "wfm", "mae", "fm", "em", "iou")):
        self.class_names = list(class_names)
        metric_names = tuple(metric_names) if metric_names else SUPPORTED
        assert set(metric_names).issubset(SUPPORTED), f"Only support: {self.suppoted_metrics}"
        self.metric_names = metric_names
        self._pending: List[tuple] = []
        self._steps: List[Dict[str, object]] = []

    def step_batch(self, logits: torch.Tensor, gts: Sequence[torch.Tensor], same_class) -> List[torch.Tensor]:
        """logits (B,1,H,W) f32 mask logits; gts: B uint8 (h_i,w_i) device tensors (ragged sizes allowed);
        same_class: B bools or a bool tensor (predicted class == ground-truth class).  Returns the uint8 masks."""
        # a device tensor of flags stays on the device until `show()` (no host synchronisation inside the loop)
        flags = same_class if isinstance(same_class, torch.Tensor) else list(same_class)
        masks = []
        for i, gt in enumerate(gts):
            h, w = int(gt.shape[-2]), int(gt.shape[-1])
            pre = mask_to_u8(logits[i:i + 1], h, w)
            self.step(pre[0], gt, flags[i] if isinstance(flags, torch.Tensor) else bool(flags[i]))
            masks.append(pre[0])
        return masks

    def step(self, pre: torch.Tensor, gt: torch.Tensor, same_class=True, gt_path: Optional[str] = None):
        """pre / gt uint8 (h,w) device tensors (ovcos_metricer.py:269-272 takes numpy arrays and two class names);
        same_class: bool, or a 0-d device tensor (read at `show()` time)."""
        assert pre.shape == gt.shape, (tuple(pre.shape), tuple(gt.shape), gt_path)
        assert pre.dtype == gt.dtype == torch.uint8, (pre.dtype, gt.dtype, gt_path)
        if not pre.is_cuda:
            raise RuntimeError("DeviceMetricer.step needs GPU tensors; there is no CPU path")
        p3, g3 = pre.reshape(1, *pre.shape[-2:]), gt.reshape(1, *gt.shape[-2:])
        stats, hist = mask_counts(p3, g3)
        wsum = mask_wfm_sums(p3, g3, hist) if "wfm" in self.metric_names else None
        flag = same_class.reshape(1).to(torch.bool) if isinstance(same_class, torch.Tensor) else bool(same_class)
        self._pending.append((stats, hist, int(pre.shape[-2]), int(pre.shape[-1]), flag, wsum))

    def _drain(self) -> None:
        if not self._pending:
            return
        stats = torch.cat([p[0] for p in self._pending]).cpu().numpy()
        hist = torch.cat([p[1] for p in self._pending]).cpu().numpy()
        wsums = torch.cat([p[5] for p in self._pending]).cpu().numpy() if "wfm" in self.metric_names else None
        dev_flags = [p[4] for p in self._pending if isinstance(p[4], torch.Tensor)]
        dev_flags = iter(torch.cat(dev_flags).cpu().tolist()) if dev_flags else iter(())
        for i, (_, _, h, w, same, _) in enumerate(self._pending):
            same = bool(next(dev_flags)) if isinstance(same, torch.Tensor) else same
            self._steps.append(metrics_from_counts(stats[i], hist[i], h, w, same, self.metric_names,
                                                   None if wsums is None else wsums[i]))
        self._pending = []

    def get_step_results(self) -> dict:
        return self._get_raw_results()

    def _get_raw_results(self) -> dict:
        self._drain()
        res: Dict[str, float] = OrderedDict()
        for m in self.metric_names:
            if m in CURVE_METRICS:
                curve = np.stack([np.asarray(s[f"{m}_curve"], dtype=np.float64) for s in self._steps]).mean(axis=0)
                res[f"adp{m}"] = np.float64(np.mean([s[f"{m}_adp"] for s in self._steps]))
                res[f"max{m}"] = np.float64(curve.max())
                res[f"avg{m}"] = np.float64(curve.mean())
            else:
                res[m] = np.float64(np.mean([s[m] for s in self._steps]))
        return res

    def show(self, num_bits: Optional[int] = 3) -> dict:
        res = self._get_raw_results()
        if isinstance(num_bits, int):
            res = {k: v.round(num_bits) for k, v in res.items()}
        return {k: float(v) for k, v in res.items()}


# ---- utils.calc_cod: Sm / Em / wFm / MAE of the float probability map (utils.py:143-165) --------------------------------------------
def cod_counts(prob: torch.Tensor, gt: torch.Tensor):
    """prob f32 (N,1,h,w) or (N,h,w) probabilities, gt uint8 (N,h,w) on the GPU -> device tensors (stats int64 (N,3), hist int32
    (N,4,2,256), moments f64 (N,4,2,2), wfm sums f64 (N,3)): everything `cod_from_counts` needs, 8.3 KB per image."""
    if prob.dim() == 4:
        prob = prob[:, 0]
    prob = prob.float().contiguous()
    gt = gt.contiguous()
    n, h, w = prob.shape
    dev = prob.device
    ws = torch.empty(hip.prob_workspace_bytes(n, h, w), dtype=torch.uint8, device=dev)
    minmax = torch.empty((n, 2), dtype=torch.float32, device=dev)
    q = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    hip.prob_quantise(prob, minmax, q, ws)
    stats, hist = mask_counts(q, gt)
    moments = torch.empty((n, 4, 2, 2), dtype=torch.float64, device=dev)
    hip.prob_moments(prob, gt, minmax, stats, ws, moments)
    wsum = torch.empty((n, 3), dtype=torch.float64, device=dev)
    hip.prob_wfm(prob, gt, minmax, _gauss49(dev), ws, wsum)
    return stats, hist, moments, wsum


def cod_from_counts(stats: np.ndarray, hist: np.ndarray, moments: np.ndarray, wsum: np.ndarray, h: int, w: int) -> Dict[str, float]:
    """One image of `calc_cod`: sm (Smeasure.cal_sm, alpha 0.5), em (mean of the 256-point E curve), wfm (beta 0.3, the class default),
    mae -- from the counters and sums of `cod_counts`, in float64 (the reference sums float32 pixels: agreement ~1e-7)."""
    hist = np.asarray(hist).astype(np.int64).reshape(4, 2, 256)
    mom = np.asarray(moments, dtype=np.float64).reshape(4, 2, 2)
    size = h * w
    total = hist.sum(axis=0)                                          # [class][level] of q = uint8(pn * 255)
    n1 = int(total[1].sum())
    n_cell = hist.sum(axis=2).astype(np.float64)                      # [quadrant][class] pixel counts
    s1, s2 = mom[:, :, 0], mom[:, :, 1]
    sum_fg, sum_bg = float(s1[:, 1].sum()), float(s1[:, 0].sum())
    mae = ((n1 - sum_fg) + sum_bg) / size                             # |pn - 1| on the foreground, |pn| on the background
    tp, fp = np.cumsum(total[1][::-1]), np.cumsum(total[0][::-1])
    em = float(np.asarray(_em(tp, fp, n1, size), dtype=np.float64).mean())
    wfm = wfm_from_sums(wsum, beta=0.3)
    with np.errstate(all="ignore"):
        mean_all = (sum_fg + sum_bg) / size
        if n1 == 0:
            sm = 1 - mean_all
        elif n1 == size:
            sm = mean_all
        else:
            def s_object(n, a, b, flip):                              # mean / ddof-1 std of pn (or 1 - pn) over one class
                x = a / n
                ssd = b - n * x * x
                x = 1 - x if flip else x
                return 2 * x / (x ** 2 + 1 + np.sqrt(max(ssd, 0.0) / (n - 1)) + _EPS)
            u = n1 / size
            obj = u * s_object(n1, sum_fg, float(s2[:, 1].sum()), False) + (1 - u) * s_object(size - n1, sum_bg, float(s2[:, 0].sum()), True)
            cnt, sx, sy = (int(v) for v in stats)
            cx, cy = int(np.round(sx / cnt)) + 1, int(np.round(sy / cnt)) + 1
            w1, w2, w3 = cx * cy / size, cy * (w - cx) / size, (h - cy) * cx / size
            reg = 0.0
            for k, wk in enumerate((w1, w2, w3, 1 - w1 - w2 - w3)):
                n = n_cell[k].sum()
                x = s1[k].sum() / n
                y = n_cell[k][1] / n
                sxx = (s2[k].sum() - n * x * x) / (n - 1)
                syy = (n_cell[k][1] * (1 - y) ** 2 + n_cell[k][0] * y ** 2) / (n - 1)
                sxy = ((s1[k][1] - n_cell[k][1] * x) * (1 - y) + (s1[k][0] - n_cell[k][0] * x) * (0 - y)) / (n - 1)
                a, b = 4 * x * y * sxy, (x ** 2 + y ** 2) * (sxx + syy)
                reg = reg + wk * (a / (b + _EPS) if a != 0 else (1 if b == 0 else 0))
            sm = max(0, 0.5 * obj + 0.5 * reg)
    return {"sm": float(sm), "em": em, "wfm": float(wfm), "mae": float(mae)}


class DeviceCod:
    """`utils.calc_cod` + the four `utils.Averager`s around it (test_ovcos_maskdecoder_edge.py:70-109): `step` queues the kernels
    for a batch of probability maps, `result()` reads the counters back once and returns the per-image means (sm, em, wfm, mae)."""

    def __init__(self):
        self._pending: List[tuple] = []

    def step(self, prob: torch.Tensor, gt: torch.Tensor) -> None:
        if not prob.is_cuda:
            raise RuntimeError("DeviceCod.step needs GPU tensors; there is no CPU path")
        self._pending.append(cod_counts(prob, gt) + (int(prob.shape[-2]), int(prob.shape[-1])))

    def result(self) -> Dict[str, float]:
        vals = []
        for stats, hist, mom, wsum, h, w in self._pending:
            st, hi, mo, wsm = stats.cpu().numpy(), hist.cpu().numpy(), mom.cpu().numpy(), wsum.cpu().numpy()
            vals += [cod_from_counts(st[i], hi[i], mo[i], wsm[i], h, w) for i in range(st.shape[0])]
        return {k: float(np.mean([v[k] for v in vals])) for k in ("sm", "em", "wfm", "mae")} if vals else {}


class DeviceClassification:
    """`Classification` (recorder/new_evaluator.py:23-100) with the counters kept on the device."""

    def __init__(self, lab2cname=None, per_class_result: bool = False, device: str = "cuda"):
        self._lab2cname = lab2cname
        self._per_class = per_class_result
        self._device = torch.device(device)
        self.reset()

    def reset(self) -> None:
        self._counters = torch.zeros(3, dtype=torch.int32, device=self._device)
        self._pred: List[torch.Tensor] = []
        self._true: List[torch.Tensor] = []

    def process(self, mo: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        """mo (B,C) scores, gt (B,) labels -> predicted class per row (int32, device)."""
        if not mo.is_cuda:
            raise RuntimeError("DeviceClassification.process needs GPU tensors; there is no CPU path")
        scores = mo.float().contiguous()
        labels = gt.to(device=mo.device, dtype=torch.int32).contiguous()
        pred = torch.empty(scores.shape[0], dtype=torch.int32, device=mo.device)
        hip.topk_accumulate(scores, labels, pred, self._counters)
        self._pred.append(pred)
        self._true.append(labels)
        return pred

    def evaluate(self) -> "OrderedDict[str, float]":
        c1, c5, total = (int(v) for v in self._counters.cpu().tolist())
        res: "OrderedDict[str, float]" = OrderedDict()
        acc = 100.0 * c1 / total
        res["accuracy"], res["error_rate"], res["top5"] = acc, 100.0 - acc, 100.0 * c5 / total
        y_true = torch.cat(self._true).cpu().numpy()
        y_pred = torch.cat(self._pred).cpu().numpy()
        f1s = []
        for c in np.unique(y_true):                      # macro F1 over the labels present (new_evaluator.py:72-77)
            tp = int(((y_pred == c) & (y_true == c)).sum())
            fp = int(((y_pred == c) & (y_true != c)).sum())
            fn = int(((y_pred != c) & (y_true == c)).sum())
            f1s.append(0.0 if 2 * tp + fp + fn == 0 else 2 * tp / (2 * tp + fp + fn))
        res["macro_f1"] = 100.0 * float(np.mean(f1s))
        if self._per_class:                              # mean of the per-class accuracies (new_evaluator.py:99-120)
            res["perclass_accuracy"] = float(np.mean([100.0 * float((y_pred[y_true == c] == c).mean())
                                                      for c in np.unique(y_true)]))
        return res
