"""GPU probe (VERDICT r5 "What's weak" #2): the lo plane of ONE attention operand zeroed IN MEMORY in front of every ViT-H attention
launch of the product library -- the data-side twin of the compile-time probe builds of tools/ab_attn_terms.sh (CVLM_ATTN_TERMS), whose
K and V rows (mask error 2.0) contradict the arithmetic (tools/precision_emulate_attn.py: K 3e-5).  With the plane zeroed the three-term
kernels compute exactly `fp16(k) . q`: if THIS measures ~1e-4 the probe builds mis-executed; if it measures 2.0 the effect is in the data.

    python tools/probe_kv_lo.py [--precision mx33] [--modes none,k,v,q,kv]      (demo geometry, B = 8, the 16 digest images)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from camouflaged_vlm_amd import digest, hip, spec, synth      # noqa: E402
from camouflaged_vlm_amd.engine import Cascade, Precision     # noqa: E402

OPS = {"q": 0, "k": 1, "v": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="mx33")
    ap.add_argument("--modes", default="none,k,v,q,kv")
    ap.add_argument("--batches", type=int, default=2)
    args = ap.parse_args()
    g, c = spec.DEMO_SAM, spec.DEMO_CLIP
    dev = torch.device("cuda:0")
    dg = digest.load(digest.golden_path("demo_digest.npz"))
    bank = torch.from_numpy(digest.load(digest.golden_path("ovcamo_constants.npz"))["bank_test"]).float()
    sd = {k: torch.from_numpy(v) for k, v in synth.make_full_state_dict(g, c).items()}
    cas = Cascade(sd, g, c, dev, Precision.named(args.precision))
    del sd
    cas.clip.set_text_bank(cas.clip.text_features(dg["eot_test"].tolist(), "test"), bank, "test")
    orig = hip.attention
    state = {"ops": (), "pads": {}}

    def patched(qkv, o, Bn, S, heads, hd, **kw):
        if kw.get("mode", 0) in (1, 2) and state["ops"]:
            assert kw.get("head_major"), "the probe assumes the head-major [3][B][H][S][hd] planes"
            lo = qkv.lo.reshape(3, -1)
            for op in state["ops"]:
                lo[op].zero_()
            pad = kw.get("pad")
            if pad is not None:                                       # pad tokens: the qkv bias row, same treatment
                key = (id(pad), state["ops"])
                if key not in state["pads"]:
                    p2 = hip.H2(pad.t.clone())
                    pl = p2.lo.reshape(3, -1)
                    for op in state["ops"]:
                        pl[op].zero_()
                    state["pads"][key] = p2
                kw["pad"] = state["pads"][key]
        return orig(qkv, o, Bn, S, heads, hd, **kw)

    hip.attention = patched
    print("# precision %s, demo geometry, B = 8, %d digest images; lo plane of the named operand(s) zeroed in memory before every ViT-H attention launch"
          % (args.precision, 8 * args.batches))
    print("# %-8s %12s %12s %10s %6s" % ("zeroed", "mask", "class logits", "IoU", "pred"))
    for mode in args.modes.split(","):
        state["ops"] = () if mode == "none" else tuple(OPS[ch] for ch in mode)
        worst = {"mask": 0.0, "logit": 0.0, "iou": 1.0, "pred": True}
        for k in range(args.batches):
            ids = list(range(8 * k, 8 * k + 8))
            per = [synth.make_inputs(g, c, batch=1, index0=i) for i in ids]
            inp, ci, cm = (torch.from_numpy(np.concatenate([p[j] for p in per])).to(dev) for j in range(3))
            m, p, l = cas.cascade(inp, ci, cm)
            r = digest.check_cascade(m, p, l, dg, ids)
            worst["mask"] = max(worst["mask"], r["max_abs_mask_err"]); worst["logit"] = max(worst["logit"], r["max_abs_class_logit_err"])
            worst["iou"] = min(worst["iou"], r["min_iou"]); worst["pred"] = worst["pred"] and r["pred_equal"]
            for kk, vv in r["max_abs_mask_err_by_set"].items():
                worst["set_" + kk] = max(worst.get("set_" + kk, 0.0), vv)
        print("  %-8s %12.3e %12.3e %10.6f %6s   %s" % (mode, worst["mask"], worst["logit"], worst["iou"], "same" if worst["pred"] else "DIFF",
                                                     " ".join("%s %.3e" % (kk[4:], vv) for kk, vv in worst.items() if kk.startswith("set_"))), flush=True)
    hip.attention = orig


if __name__ == "__main__":
    main()
