"""K-part tails of the 256^2 GEMM (CVLM_GEMM_TAIL) at the shapes that have a partial last round (B = 8), in one process."""
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
for name, M, N, K in (("proj", 32768, 1280, 1280), ("lin2", 32768, 1280, 5184), ("clip pj", 4648, 1024, 4096), ("clip fc", 4648, 4096, 1024),
                      ("clip out", 4648, 1024, 1024), ("patch", 32768, 1280, 768)):
    sc = torch.tensor([1.0, 2.0 ** -11], device="cuda").view(2, 1, 1).half()
    a = hip.H2(torch.randn(2, M, K, device="cuda").half() * sc)
    w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half() * sc)
    oh = hip.H2.empty(M, N)
    line, ref = [], None
    for rep in range(2):
        for tail in ("0", "1"):
            os.environ["CVLM_GEMM_TAIL"] = tail
            os.environ["CVLM_GEMM_VARIANT"] = "7"
            t = run(lambda: hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws))
            got = oh.float()
            if ref is None: ref = got.clone()
            err = float((got - ref).abs().max() / ref.abs().max())
            line.append(f"tail={tail}: {t:7.1f} us (rel diff {err:.1e})")
    os.environ["CVLM_GEMM_VARIANT"] = "0"
    t = run(lambda: hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws))
    print(f"{name:9s} {M}x{N}x{K}: " + "  ".join(line) + f"  | auto: {t:7.1f} us; hand-off errors {hip.gemm_workspace_errors(ws)}", flush=True)
