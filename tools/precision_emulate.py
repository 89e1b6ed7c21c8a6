"""CPU study (no GPU): what the mask / class logits lose when the CORRECTION products of the split GEMM run in narrower formats.

The exact mode forms x.w as xh.wh + xl.wh + xh.wl with four fp16 planes.  The two correction products are 2^-11 of
the main one, so they need only a few significant bits: gfx950's block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) multiplies e4m3
operands at 2x and e2m3 operands at 4x the fp16 rate.  This script emulates those operand formats inside the oracle's Linear layers
(F.linear patched in oracle/cvlm_oracle.py's namespace; accumulation in float64 so that only the operand formats differ) on the tiny
cascade and prints the error of every mode against the unpatched fp32 oracle.

    python tools/precision_emulate.py [--attn]      # --attn: the attention products (QK^T, PV) too
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from camouflaged_vlm_amd import spec, synth      # noqa: E402
from oracle import cvlm_oracle as O               # noqa: E402


def f16(v):
    return v.to(torch.float16).to(torch.float64)


def _blocks(v, n=32):
    k = v.shape[-1]
    pad = (-k) % n
    if pad:
        v = F.pad(v, (0, pad))
    return v.reshape(*v.shape[:-1], -1, n), k


def q_e4m3(v, block=32):
    """OCP e4m3 with one power-of-two scale per `block` k-elements (E8M0), as the scaled MFMA takes it."""
    b, k = _blocks(v.to(torch.float64), block)
    mx = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    s = torch.exp2(torch.ceil(torch.log2(mx / 448.0)))
    q = (b / s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) * s
    return q.reshape(*v.shape[:-1], -1)[..., :k]


def q_e2m3(v, block=32):
    """OCP fp6 e2m3 (max 7.5, subnormal step 0.125) with one power-of-two scale per block."""
    b, k = _blocks(v.to(torch.float64), block)
    mx = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    s = torch.exp2(torch.ceil(torch.log2(mx / 7.5)))
    a = (b / s)
    e = torch.floor(torch.log2(a.abs().clamp_min(1e-300))).clamp(0, 2)
    step = torch.exp2(e - 3)
    q = (torch.round(a / step) * step).clamp(-7.5, 7.5) * s
    return q.reshape(*v.shape[:-1], -1)[..., :k]


def q_e3m2(v, block=32):
    """OCP bf6 e3m2 (max 28, 2 mantissa bits)."""
    b, k = _blocks(v.to(torch.float64), block)
    mx = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    s = torch.exp2(torch.ceil(torch.log2(mx / 28.0)))
    a = (b / s)
    e = torch.floor(torch.log2(a.abs().clamp_min(1e-300))).clamp(-2, 4)
    step = torch.exp2(e - 2)
    q = (torch.round(a / step) * step).clamp(-28, 28) * s
    return q.reshape(*v.shape[:-1], -1)[..., :k]


QS = {"8": q_e4m3, "6": q_e2m3, "b6": q_e3m2, "16": f16}


def mx_planes(v, block=32):
    """The operand format of the product's `mx` mode (include/cvlm.h, cvlm_gemm_args.mx): hi = fp16(v); per 32 k-elements ONE exponent
    taken from the block's largest |hi| (E = floor(log2 max) - 7, so that every hi / 2^E < 256), hi8 = e4m3(hi / 2^E),
    lo8 = e4m3((v - hi) / 2^(E - 11)).  Returns (hi, hi8, lo8) as float64 values."""
    v = v.to(torch.float64)
    hi = f16(v)
    lo = v - hi
    hb, k = _blocks(hi, block)
    lb, _ = _blocks(lo, block)
    mx = hb.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(mx.clamp_min(2.0 ** -24))) - 7
    s_hi, s_lo = torch.exp2(e), torch.exp2(e - 11)
    q = lambda t, s: ((t / s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) * s).reshape(*v.shape[:-1], -1)[..., :k]
    return hi, q(hb, s_hi), q(lb, s_lo)


def contract(x, w, mode):
    """x [..., K] . w [N, K]^T under the operand formats of `mode`."""
    x = x.to(torch.float64)
    w = w.to(torch.float64)
    if mode == "f32":
        return x @ w.t()
    xh, wh = f16(x), f16(w)
    xl, wl = f16(x - xh), f16(w - wh)
    if mode == "fast":
        return xh @ wh.t()
    if mode == "exact3":
        return xh @ wh.t() + xl @ wh.t() + xh @ wl.t()
    if mode == "two_x":                                  # drop hi.lo_w
        return xh @ wh.t() + xl @ wh.t()
    if mode == "two_w":
        return xh @ wh.t() + xh @ wl.t()
    if mode in ("mx", "mx64"):                           # mx64: ONE exponent per 64 k-elements (round 6 study: what a shared exponent would cost)
        blk = 32 if mode == "mx" else 64
        _, xh8, xl8 = mx_planes(x, blk)
        _, wh8, wl8 = mx_planes(w, blk)
        return xh @ wh.t() + xl8 @ wh8.t() + xh8 @ wl8.t()
    if mode.startswith("mix"):                           # mix<a>_<b>: xl.wh in format a, xh.wl in format b
        a, b = mode[3:].split("_")
        qa, qb = QS[a], QS[b]
        return xh @ wh.t() + qa(xl) @ qa(wh).t() + qb(xh) @ qb(wl).t()
    raise ValueError(mode)


class PatchedF(types.ModuleType):
    """torch.nn.functional with `linear` emulated (the CLIP towers' nn.MultiheadAttention projections call it directly)."""

    def __init__(self, mode):
        super().__init__("F")
        self.__dict__.update(F.__dict__)

        def linear(x, w, b=None):
            y = contract(x, w, mode).to(x.dtype)
            return y if b is None else y + b
        self.__dict__["linear"] = linear


FAMILIES = {                       # name -> (regex on the state_dict prefix the oracle's linear() is called with, F.linear sites too?)
    "sam_qkv": (r"^image_encoder\.blocks\.\d+\.attn\.qkv$", False),
    "sam_proj": (r"^image_encoder\.blocks\.\d+\.attn\.proj$", False),
    "sam_lin1": (r"^image_encoder\.blocks\.\d+\.mlp\.lin1$", False),
    "sam_lin2": (r"^image_encoder\.blocks\.\d+\.mlp\.lin2$", False),
    "clip_mlp": (r"^clip_model\.image_encoder\..*\.mlp\.c_(fc|proj)$", False),
    "clip_attn": (r"^$", True),                                          # in_proj / out_proj of both towers' nn.MultiheadAttention
    "rest": (r"^(?!image_encoder\.blocks\.\d+\.(attn\.(qkv|proj)|mlp\.lin[12])$)(?!clip_model\.image_encoder\..*\.mlp\.c_(fc|proj)$)", False),
}


_SD = {}


def run(mode, gold, n_img=2, families=None, demo_image=None):
    g, c = (spec.TINY_SAM, spec.TINY_CLIP) if demo_image is None else (spec.DEMO_SAM, spec.DEMO_CLIP)
    if g not in _SD:
        np_sd = synth.make_full_state_dict(g, c)
        if os.environ.get("EMULATE_OUTLIERS") == "1":      # synth.apply_outliers weights: compare with tests/golden/demo_digest_outliers.npz
            np_sd = synth.apply_outliers(np_sd)
        _SD[g] = O.to_torch_sd(np_sd)
    sd = _SD[g]
    if demo_image is None:
        inp, ci, cm = (torch.from_numpy(t) for t in synth.make_inputs(g, c, n_img))
    else:
        inp, ci, cm = (torch.from_numpy(t) for t in synth.make_inputs(g, c, batch=1, index0=demo_image))
    bank = torch.from_numpy(gold["bank_test"])
    saved = O.F
    with torch.no_grad():
        tf = O.clip_text_features(sd, c, gold["eot_test"].tolist())   # the text bank is computed once, exact, in the product too
        saved_linear = O.linear
        if mode is not None and families is None:
            O.F = PatchedF(mode)
        elif mode is not None:
            import re
            pats = [re.compile(FAMILIES[f][0]) for f in families if FAMILIES[f][0] != "^$"]
            if any(FAMILIES[f][1] for f in families):
                O.F = PatchedF(mode)                                     # F.linear sites (and every O.linear below goes through F too)

            def linear(x, sd_, p):
                w, b = sd_[p + ".weight"], sd_.get(p + ".bias")
                if any(r.search(p) for r in pats):
                    y = contract(x, w, mode).to(x.dtype)
                    return y if b is None else y + b
                return F.linear(x, w, b)
            O.linear = linear
        try:
            m, pred, logits = O.cascade(inp, ci, cm, sd, g, c, tf, bank, {})
        finally:
            O.F = saved
            O.linear = saved_linear
    return m.numpy(), pred.numpy(), logits.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="f32,exact3,fast,two_x,two_w,mix8_8,mix6_6,mix16_8,mix8_16,mix16_6,mix6_16,mixb6_b6")
    ap.add_argument("--families", default="", help="comma list of family sets, each a + joined list of " + ", ".join(FAMILIES) + "; empty: every Linear")
    ap.add_argument("--demo-images", default="", help="comma list of digest image ids: run at the FULL demo geometry (minutes per image and "
                    "mode) and compare with the reference's own outputs (tests/golden/demo_digest.npz)")
    args = ap.parse_args()
    if args.demo_images:
        return demo(args)
    with np.load(os.path.join(ROOT, "tests", "golden", "tiny_cascade.npz")) as z:
        gold = {k: z[k] for k in z.files}
    m0, p0, l0 = run(None, gold)
    print("unpatched oracle vs reference golden: mask %.2e  logits %.2e" % (
        np.abs(m0 - gold["mask_logits"]).max(), np.abs(l0 - gold["class_logits"]).max()), flush=True)
    print("%-10s %12s %12s %10s %6s   (Linear layers only; against the unpatched fp32 oracle)" % ("mode", "mask max", "mask rms", "logits", "pred"))
    sets = [f.split("+") for f in args.families.split(",")] if args.families else [None]
    for mode, fam in [(m_, f_) for f_ in sets for m_ in args.modes.split(",")]:
        m, p, l = run(mode, gold, families=fam)
        if fam is not None:
            mode = mode + ":" + "+".join(fam)
        d = (m.astype(np.float64) - m0)
        print("%-28s %12.3e %12.3e %10.3e %6s   IoU %.6f" % (mode, np.abs(d).max(), np.sqrt((d * d).mean()), np.abs(l - l0).max(),
                                                            "same" if (p == p0).all() else "DIFF", O.mask_iou(m, m0)), flush=True)


def demo(args):
    import time
    from camouflaged_vlm_amd import digest
    dg = digest.load(digest.golden_path("demo_digest_outliers.npz" if os.environ.get("EMULATE_OUTLIERS") == "1" else "demo_digest.npz"))
    consts = {"bank_test": dg["bank_test"] if "bank_test" in dg else None, "eot_test": dg["eot_test"]}
    if consts["bank_test"] is None:
        with np.load(os.path.join(ROOT, "tests", "golden", "ovcamo_constants.npz")) as z:
            consts["bank_test"] = z["bank_test"]
    sets = [f.split("+") for f in args.families.split(",")] if args.families else [None]
    print("%-40s %6s %12s %12s %6s %10s   (against the REFERENCE's outputs, demo geometry)" % ("mode", "image", "mask max", "logits", "pred", "IoU"), flush=True)
    for iid in [int(i) for i in args.demo_images.split(",")]:
        for mode, fam in [(m_, f_) for f_ in sets for m_ in args.modes.split(",")]:
            t0 = time.time()
            m, p, l = run(None if mode == "oracle" else mode, consts, families=fam, demo_image=iid)
            r = digest.check_cascade(torch.from_numpy(m), torch.from_numpy(p), torch.from_numpy(l), dg, [iid])
            print("%-40s %6d %12.3e %12.3e %6s %10.6f   %.0f s" % (mode + (":" + "+".join(fam) if fam else ""), iid, r["max_abs_mask_err"],
                  r["max_abs_class_logit_err"], "same" if r["pred_equal"] else "DIFF", r["min_iou"], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
