"""How small is the lo plane of the head-major q / k / v the fold-form qkv GEMM writes (split-3 and mx operands)?  |lo| <= 2^-11 |hi| is what
dropping a lo plane from an attention product (tools/ab_attn_terms.sh) assumes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip as H
H.load()
dev = "cuda"
M, N, K = 8192, 3840, 1280
torch.manual_seed(0)
ap = H.H2.pack(torch.randn(M, K) * 0.25) if os.environ.get("PACK", "1") == "1" else H.H2(torch.stack([(torch.randn(M, K) * 0.25).half(), (torch.randn(M, K) * 1e-4).half()]))
wp = H.H2.pack(torch.randn(N, K) * 0.03)
A_il = H.H2IL.from_planes(H.H2(ap.t.to(dev)))
A_mx = H.H2MX.from_planes(ap); A_mx = H.H2MX(A_mx.t.to(dev), A_mx.s.to(dev), None, A_mx.C)
W = H.H2(wp.t.to(dev)); W_il = H.interleave_planes(W)
W_mx = H.H2MX.from_planes(wp); W_mx = H.H2MX(W_mx.t.to(dev), W_mx.s.to(dev), None, W_mx.C)
ws = H.new_gemm_workspace(dev)
merged = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.01], 1).contiguous()
kw = dict(bias=torch.randn(N, device=dev), workspace=ws, w_il=W_il, ln_fold=(merged, torch.randn(N, device=dev)), head_major=(4096, 16, 80))
for mode in ("split3", "mx"):
    out = H.H2.empty(M, N, device=dev)
    k2 = dict(kw)
    if mode == "mx": k2["w_mx"] = W_mx
    H.gemm(A_il if mode == "split3" else A_mx, W, M, N, K, out_h2=out, **k2)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(out.t.float())
    if bool(bad.any()):
        idx = bad.nonzero()
        print(mode, "non-finite values:", int(bad.sum()), "first", idx[0].tolist(), "last", idx[-1].tolist(), "errors", H.gemm_workspace_errors(ws))
    hi, lo = out.t[0].float().view(3, -1), out.t[1].float().view(3, -1)
    for i, nm in enumerate("qkv"):
        r = (lo[i].abs() / hi[i].abs().clamp_min(1e-6))
        print(f"{mode:7s} {nm}: max|hi| {hi[i].abs().max():9.3f}  max|lo| {lo[i].abs().max():.3e}  max |lo|/|hi| {r.max():.3e}  rms lo/rms hi {lo[i].pow(2).mean().sqrt() / hi[i].pow(2).mean().sqrt():.3e}")
