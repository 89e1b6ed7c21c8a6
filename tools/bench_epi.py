"""Probe: per-tile fixed cost (prologue + epilogue) of the GEMM: tiny K, large M x N."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
M, N = 32768, 5120
for K in (32, 64, 128, 256):
    a = hip.H2(torch.randn(2, M, K, device="cuda").half()); w = hip.H2(torch.randn(2, N, K, device="cuda").half())
    oh = hip.H2.empty(M, N); of = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    for name, kw in (("h2 out", dict(out_h2=oh)), ("f32 out", dict(out_f32=of)), ("h2+bias+gelu", dict(out_h2=oh, bias=bias, act=1)),
                     ("f32+bias+res", dict(out_f32=of, bias=bias, residual=of))):
        for _ in range(2): hip.gemm(a, w, M, N, K, split=3, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [hip.gemm(a, w, M, N, K, split=3, **kw) for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        print(f"K={K:4d} {name:14s}: {e0.elapsed_time(e1)/5*1e3:8.1f} us", flush=True)
