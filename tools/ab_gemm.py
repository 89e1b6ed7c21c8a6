"""A/B of GEMM variants inside one process (same box, same clocks): CVLM_GEMM_VARIANT is re-read per call when
CVLM_GEMM_VARIANT_LIVE=1.  Usage: python tools/ab_gemm.py 7 37 ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip
hip.load()
variants = [int(v) for v in sys.argv[1:]] or [7]
shapes = [("sam qkv", 32768, 3840, 1280), ("sam proj", 32768, 1280, 1280), ("sam lin1", 32768, 5120, 1280), ("sam lin2", 32768, 1280, 5120)]
for name, M, N, K in shapes:
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    w = hip.H2(torch.randn(2, N, K, device="cuda").half())
    out = hip.H2.empty(M, N)
    res = {v: [] for v in variants}
    for rep in range(3):
        for v in variants:
            os.environ["CVLM_GEMM_VARIANT"] = str(v)
            hip.gemm(a, w, M, N, K, out_h2=out, split=3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                hip.gemm(a, w, M, N, K, out_h2=out, split=3)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 100)
    print(f"{name:9s} " + "  ".join(f"v{v}: {min(r):7.1f} us" for v, r in res.items()), flush=True)
