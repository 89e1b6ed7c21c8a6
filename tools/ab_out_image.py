"""Cost of the image epilogue's half-line stores: the lin1 / lin2-shaped launches (B = 8) with the h2 output (and residual) in planes vs in
the 128-byte-row image, operands staged from images in both.  Usage: python tools/ab_out_image.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
dev = "cuda"
ws = hip.new_gemm_workspace(dev)
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, M, N, K, act in (("lin1 (fold + GELU)", 32768, 5120, 1280, 1), ("lin2 (h2 residual + statistics)", 32768, 1280, 5120, 0)):
    a = hip.H2(torch.randn(2, M, K, device=dev).half() * torch.tensor([1.0, 2.0 ** -11], device=dev).view(2, 1, 1).half())
    w = hip.H2(torch.randn(2, N, K, device=dev).half() * torch.tensor([0.05, 0.05 * 2.0 ** -11], device=dev).view(2, 1, 1).half())
    ai, wi = hip.H2IL.from_planes(a), hip.interleave_planes(w)
    bias = torch.randn(N, device=dev)
    res = {}
    if act:
        st = torch.empty(hip.stats_pieces(K), M, 2, device=dev); mrg = torch.empty(M, 2, device=dev)
        hip.row_stats_split(a.float(), 1.0, hip.H2.empty(M, K), st, M, K)
        hip.ln_stats_merge(st, M, K, 1e-6, mrg, ws)
        cs = torch.randn(N, device=dev)
        op, oi = hip.H2.empty(M, N), hip.H2IL.empty(M, N)
        for rep in range(3):
            for tag, o in (("planes", op), ("image", oi)):
                res.setdefault(tag, []).append(t(lambda: hip.gemm(ai, w, M, N, K, bias=bias, act=1, out_h2=o, ln_fold=(mrg, cs), workspace=ws, w_il=wi)))
    else:
        st = torch.empty(hip.stats_pieces(N), M, 2, device=dev)
        xp = hip.H2(torch.randn(2, M, N, device=dev).half() * 0.01); xi = hip.H2IL.from_planes(xp)
        for rep in range(3):
            for tag, x in (("planes", xp), ("image", xi)):
                res.setdefault(tag, []).append(t(lambda: hip.gemm(ai, w, M, N, K, bias=bias, out_h2=x, residual_h2=(x, 1.0), out_scale=1.0, row_stats=st, workspace=ws, w_il=wi)))
    print(f"{name:34s} {M}x{N}x{K}: output in planes {min(res['planes']):7.1f} us   in the image {min(res['image']):7.1f} us", flush=True)
