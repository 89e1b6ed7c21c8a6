"""What the epilogue forms of the 256^2 GEMM cost at the SAM shapes (B = 8): h2 out, f32 out, f32 out + residual."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
ws = hip.new_gemm_workspace("cuda")
def run(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in (("proj", 32768, 1280, 1280), ("lin2", 32768, 1280, 5184), ("qkv", 32768, 3840, 1280), ("clip out", 4648, 1024, 1024), ("clip pj", 4648, 1024, 4096)):
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half())
    oh = hip.H2.empty(M, N)
    x = torch.randn(M, N, device="cuda")
    big = torch.empty(64 * 1024 * 1024, device="cuda")                  # 256 MB: flush the Infinity Cache between runs
    res = {}
    res["h2 out"] = run(lambda: hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws))
    res["f32 out"] = run(lambda: hip.gemm(a, w, M, N, K, out_f32=x, workspace=ws))
    res["f32 out + residual (in place)"] = run(lambda: hip.gemm(a, w, M, N, K, out_f32=x, residual=x, alpha=1e-3, workspace=ws))
    def cold():
        big.zero_()
        hip.gemm(a, w, M, N, K, out_f32=x, residual=x, alpha=1e-3, workspace=ws)
    t_all = run(cold, n=5)
    t_flush = run(lambda: big.zero_(), n=5)
    res["f32 + residual, caches flushed"] = t_all - t_flush
    print(f"{name:9s} {M}x{N}x{K}: " + "  ".join(f"{k}: {v:7.1f} us" for k, v in res.items()), flush=True)
