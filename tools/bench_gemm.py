"""GEMM micro-benchmark on the shapes of the cascade (B = 8).  Variants other than 0 / 1 / 2 / 7 exist only in a probe build (`make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES`).
Usage: python tools/bench_gemm.py [split] [alias]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
split = int(sys.argv[1]) if len(sys.argv) > 1 else 3
alias = len(sys.argv) > 2 and sys.argv[2] == "alias"      # lda = ldw = 0: every row aliases row 0 (all cache hits)
shapes = [("sam qkv", 32768, 3840, 1280), ("sam proj", 32768, 1280, 1280), ("sam lin1", 32768, 5120, 1280),
          ("sam lin2", 32768, 1280, 5120), ("clip in", 4648, 3072, 1024), ("clip out", 4648, 1024, 1024),
          ("clip fc", 4648, 4096, 1024), ("clip pj", 4648, 1024, 4096)]
if os.environ.get("SHAPES") == "few":
    shapes = [shapes[2], shapes[3]]
tot_f, tot_t = 0.0, 0.0
for name, M, N, K in shapes:
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    w = hip.H2(torch.randn(2, N, K, device="cuda").half())
    out = hip.H2.empty(M, N)
    kw = dict(lda=0, ldw=0) if alias else {}
    for _ in range(2):
        hip.gemm(a, w, M, N, K, out_h2=out, split=split, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        hip.gemm(a, w, M, N, K, out_h2=out, split=split, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / ms / 1e9
    tot_f += 2.0 * M * N * K; tot_t += ms
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d} split={split}: {ms*1e3:8.1f} us  {tf:7.1f} TF algorithmic  ({tf*split:7.1f} TF issued)", flush=True)
print(f"sum: {tot_f/tot_t/1e9:.1f} TF algorithmic")
