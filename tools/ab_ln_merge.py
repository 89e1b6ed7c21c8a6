"""Cost of the LayerNorm-fold statistics merge in the consuming GEMM (qkv / lin1 shapes of the ViT-H blocks at B = 8): the folded
launch against the plain launch of the same shape.  Run once per library build (CVLM_PROBE_LIB=...) on the same box for an A/B.
Usage: python tools/ab_ln_merge.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
dev = "cuda"
ws = hip.new_gemm_workspace(dev)
for name, M, N, K, act in (("qkv", 32768, 3840, 1280, 0), ("lin1", 32768, 5120, 1280, 1), ("clip fc", 9296, 4096, 1024, 2)):
    a = hip.H2(torch.randn(2, M, K, device=dev).half() * torch.tensor([1.0, 2.0 ** -11], device=dev).view(2, 1, 1).half())
    w = hip.H2(torch.randn(2, N, K, device=dev).half() * torch.tensor([0.05, 0.05 * 2.0 ** -11], device=dev).view(2, 1, 1).half())
    out = hip.H2.empty(M, N)
    st = torch.empty(hip.stats_pieces(K), M, 2, device=dev)
    x = a.float()
    hip.row_stats_split(x, 1.0, hip.H2.empty(M, K), st, M, K)
    mrg = torch.empty(M, 2, device=dev)
    hip.ln_stats_merge(st, M, K, 1e-6, mrg, ws)
    cs, bias = torch.randn(N, device=dev), torch.randn(N, device=dev)
    res = {}
    for kind in ("plain", "fold", "plain", "fold"):
        kw = dict(ln_fold=(mrg, cs)) if kind == "fold" else {}
        for _ in range(2):
            hip.gemm(a, w, M, N, K, bias=bias, act=act, out_h2=out, workspace=ws, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.gemm(a, w, M, N, K, bias=bias, act=act, out_h2=out, workspace=ws, **kw)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(kind, []).append(e0.elapsed_time(e1) * 100)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.ln_stats_merge(st, M, K, 1e-6, mrg, ws)
    e1.record(); torch.cuda.synchronize()
    print(f"         cvlm_ln_stats_merge M={M} D={K}: {e0.elapsed_time(e1) * 50:.1f} us per launch (back to back)")
    print(f"{name:8s} {M}x{N}x{K}: plain {min(res['plain']):7.1f} us   LayerNorm-folded {min(res['fold']):7.1f} us   (+{100 * (min(res['fold']) / min(res['plain']) - 1):.1f} %)", flush=True)
