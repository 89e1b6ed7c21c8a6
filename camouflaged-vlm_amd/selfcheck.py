"""smoke(): one tiny cascade through the HIP path on cuda:0, checked against the CPU oracle."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import REPO_DIR, spec, synth
from .engine import Cascade, Precision


def build_tiny(device, precision: Precision = Precision(), g=spec.TINY_SAM, c=spec.TINY_CLIP, seed: int = 0):
    sd_np = synth.make_full_state_dict(g, c, seed)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    cas = Cascade(sd, g, c, device, precision)
    return cas, sd_np


def smoke() -> None:
    assert torch.cuda.is_available(), "smoke() needs an MI355X"
    dev = torch.device("cuda:0")
    if REPO_DIR not in sys.path:
        sys.path.insert(0, REPO_DIR)
    from oracle import cvlm_oracle as O          # checker only
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    cas, sd_np = build_tiny(dev)
    eot = spec.default_eot(c, "test")
    bank = synth.make_text_bank(c.n_cls_test, c.embed_dim, "test")
    cas.clip.set_text_bank(cas.clip.text_features(eot, "test"), torch.from_numpy(bank), "test")
    inp, ci, cm = synth.make_inputs(g, c, batch=1)
    masks, pred, logits = cas.cascade(torch.from_numpy(inp).to(dev), torch.from_numpy(ci).to(dev),
                                      torch.from_numpy(cm).to(dev))
    torch.cuda.synchronize()
    sd = O.to_torch_sd(sd_np)
    with torch.no_grad():
        tf = O.clip_text_features(sd, c, eot)
        m_ref, p_ref, l_ref = O.cascade(torch.from_numpy(inp), torch.from_numpy(ci), torch.from_numpy(cm), sd, g, c,
                                        tf, torch.from_numpy(bank))
    dm = float((masks.cpu() - m_ref).abs().max())
    dl = float((logits.cpu() - l_ref).abs().max())
    iou = O.mask_iou(masks.cpu().numpy(), m_ref.numpy())
    print(f"smoke: max|mask - oracle| = {dm:.2e}, max|logits - oracle| = {dl:.2e}, IoU = {iou:.6f}, "
          f"pred {pred.cpu().tolist()} vs {p_ref.tolist()}")
    assert dm < 1e-3 and dl < 1e-3 and iou >= 0.999 and pred.cpu().tolist() == p_ref.tolist()
    # The tiny geometry (D = 160) has no mx launches (K % 64 != 0): one launch of the mx GEMM form the demo geometry's default precision
    # runs on -- mx operands in, h2 residual epilogue, mx out -- against the fp64 product of the same h2 values.
    from . import hip
    M, N, K = 512, 256, 256
    gen = torch.Generator().manual_seed(3)
    a, w, r = (torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) * K ** -0.5, torch.randn(M, N, generator=gen))
    ap, wp, rp = hip.H2.pack(a), hip.H2.pack(w), hip.H2.pack(r)
    mv = lambda m: hip.H2MX(m.t.to(dev), m.s.to(dev), None if m.lo is None else m.lo.to(dev), m.C)
    out = mv(hip.H2MX.from_planes(rp, lo_plane=True))
    hip.gemm(mv(hip.H2MX.from_planes(ap)), hip.H2(wp.t.to(dev)), M, N, K, out_h2=out, residual_h2=(out, 1.0), w_mx=mv(hip.H2MX.from_planes(wp)))
    torch.cuda.synchronize()
    want = ap.float().double() @ wp.float().double().t() + rp.float().double()
    got = out.hi().float().cpu().double() + out.lo.float().cpu().double()
    dg = float((got - want).abs().max() / want.abs().max())
    print(f"smoke: mx GEMM {M}x{N}x{K} (f16 hi.hi + e4m3 corrections) vs fp64: {dg:.2e} of the largest output")
    assert dg < 1e-4
