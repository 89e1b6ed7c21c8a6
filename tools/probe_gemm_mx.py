"""What bounds the mx GEMM kernel: the probe forms of its main loop (gemm_kernel.h MX branch, DBG), alternating in one process.
Needs the probe build:  make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES LIBDIR=../lib_probe
    CVLM_PROBE_LIB=camouflaged-vlm_amd/lib_probe/libcvlm_hip.so python tools/probe_gemm_mx.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip as H
H.load()
dev = "cuda"
M, N, K = 32768, (5120 if os.environ.get("SHAPE") == "lin1" else 3840), 1280
torch.manual_seed(0)
ap = H.H2(torch.stack([(torch.randn(M, K) * 0.25).half(), (torch.randn(M, K) * 1e-4).half()]))
wp = H.H2(torch.stack([(torch.randn(N, K) * 0.5).half(), (torch.randn(N, K) * 2e-4).half()]))
mv = lambda m: H.H2MX(m.t.to(dev), m.s.to(dev), None, m.C)
A_il, A_mx = H.H2IL.from_planes(H.H2(ap.t.to(dev))), mv(H.H2MX.from_planes(ap))
W = H.H2(wp.t.to(dev)); W_il = H.interleave_planes(W); W_mx = mv(H.H2MX.from_planes(wp))
merged = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.01], 1).contiguous()
kw = dict(bias=torch.randn(N, device=dev), workspace=H.new_gemm_workspace(dev), w_il=W_il, ln_fold=(merged, torch.randn(N, device=dev)))
if os.environ.get("SHAPE") == "lin1":                                  # lin1: GELU, mx output (the hidden rows)
    kw.update(act=H.ACT_GELU, out_scale=0.25)
    out = H.H2MX.empty(M, N, device=dev)
else:
    kw.update(head_major=(4096, 16, 80))
    out = H.H2.empty(M, N, device=dev)
forms = [("split-3 kernel", None, "0"), ("mx kernel", 1, "0"), ("mx, no DMA in the steady state", 1, "101"), ("mx, DMA only (no reads, no MFMAs)", 1, "102"),
         ("mx, no fragment reads", 1, "109"), ("mx, only the f16 units multiply", 1, "110"), ("mx, only the fp8 units multiply", 1, "111"),
         ("mx, main loop only (no epilogue)", 1, "106"), ("mx, epilogue without its global stores", 1, "103"),
         ("mx, epilogue: LDS staging only (no split / conversion / stores)", 1, "105"),
         ("mx, epilogue stores aimed at 128 KB that stay in L2 (same instructions)", 1, "107"), ("mx, DMA only, sc0", 1, "112"), ("mx, DMA only, nt", 1, "113"), ("mx, DMA only, sc1", 1, "114"),
         ("mx, DMA only, sc0 sc1", 1, "115")]
res = {f[0]: [] for f in forms}
for rep in range(3):
    for name, mx, var in forms:
        os.environ["CVLM_GEMM_VARIANT"] = var
        a = A_mx if mx else A_il
        k2 = dict(kw, w_mx=W_mx) if mx else kw
        H.gemm(a, W, M, N, K, out_h2=out, **k2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            H.gemm(a, W, M, N, K, out_h2=out, **k2)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) * 100)
print(f"{os.environ.get('SHAPE', 'qkv')} {M}x{N}x{K} (LayerNorm fold; qkv: head-major h2 store, lin1: GELU + mx store), us per launch:")
for name, r in res.items():
    print(f"  {name:66s} {min(r):7.1f}")
