"""A/B of the persistent 256^2 GEMM (CVLM_GEMM_PERSIST) at the encoder shapes, inside one process (B = 8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip
hip.load()
ws = hip.new_gemm_workspace("cuda")
def run(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in (("qkv", 32768, 3840, 1280), ("lin1", 32768, 5120, 1280), ("proj", 32768, 1280, 1280), ("lin2", 32768, 1280, 5184),
                      ("clip fc", 4648, 4096, 1024), ("clip qkv", 4648, 3072, 1024)):
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half())
    oh = hip.H2.empty(M, N)
    ref = None
    line = []
    for rep in range(2):
        for pv in ("0", "1"):
            os.environ["CVLM_GEMM_PERSIST"] = pv
            t = run(lambda: hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws))
            got = oh.t.float().sum(0)
            if ref is None: ref = got.clone()
            same = bool((got == ref).all())
            line.append(f"persist={pv}: {t:7.1f} us{'' if same else ' MISMATCH'}")
    print(f"{name:9s} {M}x{N}x{K}: " + "  ".join(line), flush=True)
