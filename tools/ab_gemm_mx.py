"""A/B of the mx operand form against the split-3 form of the same launch (include/cvlm.h ABI 10), alternating inside one process:
the three ViT-H block GEMMs that run on mx operands (qkv: LayerNorm fold, head-major store; lin1: LayerNorm fold + GELU, mx out;
lin2: h2 residual + row statistics, mx out + lo plane), at B = 8 (M = 32768) or SHAPES=b1 (M = 4096) / clip2 (c_proj of the fused
CLIP forward).  Prints microseconds per launch and algorithmic TFLOP/s.
    python tools/ab_gemm_mx.py            # PROBE=1: launches of the probe library named by CVLM_PROBE_LIB instead
"""
import os, sys, time, torch
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip as H
H.load()
dev = "cuda"
M = {"b1": 4096}.get(os.environ.get("SHAPES", ""), 32768)
shapes = [("qkv", M, 3840, 1280, "fold_hm"), ("lin1", M, 5120, 1280, "fold_gelu"), ("lin2", M, 1280, 5184, "h2res")]
if os.environ.get("SHAPES") == "clip2":
    shapes = [("clip pj", 9296, 1024, 4096, "h2res")]
ws = H.new_gemm_workspace(dev)
REPS, INNER = int(os.environ.get("REPS", "3")), int(os.environ.get("INNER", "10"))
for name, M, N, K, form in shapes:
    torch.manual_seed(0)
    ap, wp = H.H2.pack(torch.randn(M, K) * 0.25), H.H2.pack(torch.randn(N, K) * 0.5)      # real split planes: an mx image of planes whose lo is not the
    #                                                                                        rounding residual of hi overflows e4m3 (NaN bytes)
    A_il, A_mx = H.H2IL.from_planes(H.H2(ap.t.to(dev))), H.H2MX.from_planes(ap)
    A_mx = H.H2MX(A_mx.t.to(dev), A_mx.s.to(dev), None, A_mx.C)
    W = H.H2(wp.t.to(dev))
    W_il = H.interleave_planes(W)
    W_mx = H.H2MX.from_planes(wp)
    W_mx = H.H2MX(W_mx.t.to(dev), W_mx.s.to(dev), None, W_mx.C)
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias, workspace=ws, w_il=W_il)
    if form.startswith("fold"):
        merged = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.01], 1).contiguous()
        kw.update(ln_fold=(merged, torch.randn(N, device=dev)))
    if form == "fold_hm":
        kw.update(head_major=(4096, 16, 80))
        outs = {"split3": H.H2.empty(M, N, device=dev), "mx": H.H2.empty(M, N, device=dev)}
    elif form == "fold_gelu":
        kw.update(act=H.ACT_GELU, out_scale=0.25)
        outs = {"split3": H.H2IL.empty(M, N, device=dev), "mx": H.H2MX.empty(M, N, device=dev)}
    else:
        r_il = H.H2IL.from_planes(H.H2(H.H2.pack(torch.randn(M, N)).t.to(dev)))
        r_mx = H.H2MX.empty(M, N, device=dev, lo_plane=True)
        r_mx.t.copy_(H.H2MX.from_planes(H.H2.pack(torch.randn(M, N))).t)
        r_mx.lo.normal_(std=1e-4)
        stats = torch.zeros(H.stats_pieces(N), M, 2, device=dev)
        kw.update(row_stats=stats)
        outs = {"split3": r_il, "mx": r_mx}
    gms = [int(x) for x in os.environ.get("GROUP_MS", "").split(",") if x]          # CVLM_GEMM_GROUP_M values to race (tile rows per L2 super-tile)
    modes = ["split3", "mx"] + ["mx gm %d" % k for k in gms]
    outs.update({m: outs["mx"] for m in modes[2:]})
    res = {m: [] for m in modes}
    for rep in range(REPS):
        for mode in modes:
            os.environ["CVLM_GEMM_GROUP_M"] = mode.split()[-1] if " gm " in mode else "0"
            a = A_il if mode == "split3" else A_mx
            k2 = dict(kw)
            if mode != "split3":
                k2["w_mx"] = W_mx
            if form == "h2res":
                k2["residual_h2"] = (outs[mode], 1.0)
            H.gemm(a, W, M, N, K, out_h2=outs[mode], **k2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(INNER):
                H.gemm(a, W, M, N, K, out_h2=outs[mode], **k2)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) * 1e3 / INNER)
    fl = 2.0 * M * N * K
    print(f"{name:8s} {M}x{N}x{K} " + "  ".join(f"{m}: {min(r):7.1f} us = {fl / min(r) / 1e6:5.0f} TF/s" for m, r in res.items()) +
          f"   ({min(res['split3']) / min(res['mx']):.3f} x)", flush=True)
assert H.gemm_workspace_errors(ws) == 0
