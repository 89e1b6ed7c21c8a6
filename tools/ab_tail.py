import os, sys, torch
sys.path.insert(0, "/root/repo")
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
for name, M, N, K in (("proj", 32768, 1280, 1280), ("lin2", 32768, 1280, 5184), ("qkv", 32768, 3840, 1280), ("lin1", 32768, 5120, 1280)):
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half())
    oh = hip.H2.empty(M, N)
    for rep in range(2):
        t1 = run(lambda: hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws))
        t0 = run(lambda: hip.gemm(a, w, M, N, K, out_h2=oh))
        print(f"{name:5s}: tail split {t1:7.1f} us   whole tiles {t0:7.1f} us", flush=True)
print(hip.gemm_workspace_errors(ws))
