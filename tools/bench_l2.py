"""Probe: every workgroup streams the SAME operands (batch stride 0) -> pure L2-hit DMA rate of the GEMM loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
M = N = int(os.environ.get("TILE", "256")); K = 5120; batch = 2048
a = hip.H2(torch.randn(2, M, K, device="cuda").half()); w = hip.H2(torch.randn(2, N, K, device="cuda").half())
out = hip.H2.empty(M, N)
def f(): hip.gemm(a, w, M, N, K, out_h2=out, split=3, batch=batch, stride_a=0, stride_w=0, stride_oh=0)
for _ in range(2): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [f() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
byts = batch * (K // 32) * (M + N) * 64 * 2
print(f"variant {os.environ.get('CVLM_GEMM_VARIANT')} tile {M}: {ms*1e3:.1f} us, DMA {byts/ms/1e9:.2f} TB/s, {2.0*M*N*K*batch*3/ms/1e9:.0f} TF issued")
