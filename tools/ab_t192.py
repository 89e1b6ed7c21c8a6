"""192 x 256 tiles against 256 x 256 tiles on the producer-side shapes of the fused CLIP forward (h2 residual + piece statistics),
in one process (CVLM_GEMM_VARIANT_LIVE).  Usage: python tools/ab_t192.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip
hip.load()
dev = "cuda"
ws = hip.new_gemm_workspace(dev)
pl = torch.tensor([1.0, 2.0 ** -11], device=dev).view(2, 1, 1).half()
for name, M, N, K in (("clip out (16 img)", 9296, 1024, 1024), ("clip pj (16 img)", 9296, 1024, 4096), ("clip out (8 img)", 4648, 1024, 1024),
                      ("clip pj (8 img)", 4648, 1024, 4096)):
    a = hip.H2(torch.randn(2, M, K, device=dev).half() * pl)
    w = hip.H2(torch.randn(2, N, K, device=dev).half() * pl * 0.05)
    x = hip.H2(torch.randn(2, M, N, device=dev).half() * pl)
    st = torch.empty(hip.stats_pieces(N), M, 2, device=dev)
    bias = torch.randn(N, device=dev)
    res = {"0": [], "1": []}
    for rep in range(3):
        for flag in ("0", "1"):
            os.environ["CVLM_GEMM_T192"] = flag
            for _ in range(2):
                hip.gemm(a, w, M, N, K, bias=bias, out_h2=x, residual_h2=(x, 4.0), out_scale=0.25, row_stats=st, workspace=ws)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                hip.gemm(a, w, M, N, K, bias=bias, out_h2=x, residual_h2=(x, 4.0), out_scale=0.25, row_stats=st, workspace=ws)
            e1.record(); torch.cuda.synchronize()
            res[flag].append(e0.elapsed_time(e1) * 100)
    print(f"{name:18s} {M}x{N}x{K}: 256-row tiles {min(res['0']):7.1f} us   192-row tiles {min(res['1']):7.1f} us", flush=True)
