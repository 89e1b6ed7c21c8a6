"""What bounds the mx GEMM kernel: the probe forms of its main loop (gemm_kernel.h MX branch, DBG), alternating in one process.
Needs the probe build:  make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES LIBDIR=../lib_probe
    CVLM_PROBE_LIB=camouflaged-vlm_amd/lib_probe/libcvlm_hip.so python tools/probe_gemm_mx.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip as H
H.load()
dev = "cuda"
SHAPE = os.environ.get("SHAPE", "qkv")
M, N, K = {"qkv": (32768, 3840, 1280), "lin1": (32768, 5120, 1280), "lin2": (32768, 1280, 5184)}[SHAPE]
torch.manual_seed(0)
ap, wp = H.H2.pack(torch.randn(M, K) * 0.25), H.H2.pack(torch.randn(N, K) * 0.5)       # real split planes (an mx image of unrelated planes holds NaN bytes)
mv = lambda m: H.H2MX(m.t.to(dev), m.s.to(dev), None if m.lo is None else m.lo.to(dev), m.C)
A_il, A_mx = H.H2IL.from_planes(H.H2(ap.t.to(dev))), mv(H.H2MX.from_planes(ap))
W = H.H2(wp.t.to(dev)); W_il = H.interleave_planes(W); W_mx = mv(H.H2MX.from_planes(wp))
kw = dict(bias=torch.randn(N, device=dev), workspace=H.new_gemm_workspace(dev), w_il=W_il)
out_s3 = None
if SHAPE == "lin2":                                                    # lin2: h2 residual + row statistics, mx out + lo plane, K-parts of the last half round
    res = H.H2.pack(torch.randn(M, N))
    out = mv(H.H2MX.from_planes(res, lo_plane=True))
    out_s3 = H.H2IL.from_planes(H.H2(res.t.to(dev)))
    kw.update(row_stats=torch.zeros(H.stats_pieces(N), M, 2, device=dev))
else:
    merged = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.01], 1).contiguous()
    kw.update(ln_fold=(merged, torch.randn(N, device=dev)))
    if SHAPE == "lin1":                                                # lin1: GELU, mx output (the hidden rows)
        kw.update(act=H.ACT_GELU, out_scale=0.25)
        out = H.H2MX.empty(M, N, device=dev)
    else:
        kw.update(head_major=(4096, 16, 80))
        out = H.H2.empty(M, N, device=dev)
if SHAPE == "lin2":
    PROBES = {"101", "102", "103", "106", "116", "117", "118", "119", "120", "121", "122"}
forms = [("split-3 kernel", None, "0"), ("mx kernel", 1, "0"), ("mx, no DMA in the steady state", 1, "101"), ("mx, DMA only (no reads, no MFMAs)", 1, "102"),
         ("mx, no fragment reads", 1, "109"), ("mx, only the f16 units multiply", 1, "110"), ("mx, only the fp8 units multiply", 1, "111"),
         ("mx, main loop only (no epilogue)", 1, "106"), ("mx, epilogue without its global stores", 1, "103"),
         ("mx, epilogue: LDS staging only (no split / conversion / stores)", 1, "105"),
         ("mx, epilogue stores aimed at 128 KB that stay in L2 (same instructions)", 1, "107"), ("mx, DMA only, sc0", 1, "112"), ("mx, DMA only, nt", 1, "113"), ("mx, DMA only, sc1", 1, "114"),
         ("mx, DMA only, sc0 sc1", 1, "115"),
         ("mx, DMA only, activation rows wrapped into rows 0..255 (L2-resident)", 1, "116"),
         ("mx, DMA only, activation rows wrapped into rows 0..8191 (infinity-cache resident)", 1, "117"),
         ("mx, DMA only, activation AND weight rows wrapped into rows 0..255", 1, "118"),
         ("mx kernel, activation rows wrapped into rows 0..255 (timing only)", 1, "119"),
         ("3-byte operand, stream only: e4m3 units fetch half their lines (timing only)", 1, "120"),
         ("3-byte operand: half lines + hi8 formed in registers (timing only)", 1, "121"),
         ("hi8 formed in registers, full 4-byte stream (timing only)", 1, "122")]
if os.environ.get("ONLY3"):
    forms = [f for f in forms if f[2] in ("0", "102", "120", "121", "122")]
if SHAPE == "lin2":
    forms = [f for f in forms if f[2] == "0" or f[2] in PROBES]
res = {f[0]: [] for f in forms}
for rep in range(3):
    for name, mx, var in forms:
        os.environ["CVLM_GEMM_VARIANT"] = var
        a = A_mx if mx else A_il
        k2 = dict(kw, w_mx=W_mx) if mx else dict(kw)
        o = out if (mx or out_s3 is None) else out_s3
        if SHAPE == "lin2":
            k2["residual_h2"] = (o, 1.0)
        H.gemm(a, W, M, N, K, out_h2=o, **k2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            H.gemm(a, W, M, N, K, out_h2=o, **k2)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) * 100)
print(f"{SHAPE} {M}x{N}x{K} (qkv: LayerNorm fold, head-major h2 store; lin1: fold + GELU, mx store; lin2: h2 residual + statistics, mx store + lo plane), us per launch:")
for name, r in res.items():
    print(f"  {name:66s} {min(r):7.1f}")
