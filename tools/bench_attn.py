"""Attention micro-benchmark at the cascade's shapes (B = 8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip
hip.load()
split = (3, 3) if len(sys.argv) < 2 else tuple(int(c) for c in sys.argv[1])
hm = os.environ.get("HM", "1") == "1"
SCALE = float(os.environ["SCALE"]) if "SCALE" in os.environ else None      # SCALE=1: the factor folded into the projection (kernels skip the re-split)
def run(name, fn, flops, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:14s} split={split}: {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF algorithmic", flush=True)
B, H, hd, G = 8, 16, 80, 64
D, S = H * hd, G * G
qkv = hip.H2(torch.randn(2, B * S, 3 * D, device="cuda").half())
out = hip.H2.empty(B * S, D)
rg = hip.H2((torch.randn(2, 2 * G - 1, hd, device="cuda") * 0.1).half())
rw = hip.H2((torch.randn(2, 27, hd, device="cuda") * 0.1).half())
pad = hip.H2((torch.randn(2, 3 * D, device="cuda") * 0.1).half())
run("sam global", lambda: hip.attention(qkv, out, B, S, H, hd, mode=1, grid=G, rel_h=rg, rel_w=rg, split_qk=split[0], split_pv=split[1], head_major=hm),
    4.0 * B * H * S * S * hd)
run("sam window", lambda: hip.attention(qkv, out, B, S, H, hd, mode=2, grid=G, window=14, pad=pad, rel_h=rw, rel_w=rw, split_qk=split[0], split_pv=split[1], head_major=hm, scale=SCALE),
    4.0 * B * H * 25 * 196 * 196 * hd)
Hc, hc, Sc = 16, 64, 581
q2 = hip.H2(torch.randn(2, B * Sc, 3 * Hc * hc, device="cuda").half())
o2 = hip.H2.empty(B * Sc, Hc * hc)
run("clip vision", lambda: hip.attention(q2, o2, B, Sc, Hc, hc, mode=0, split_qk=split[0], split_pv=split[1]), 4.0 * B * Hc * Sc * Sc * hc)
if os.environ.get("HIRES", "1") == "1":                                  # 1536^2 ViT-H (BASELINE configs[4]): 96 x 96 map, B = 4
    B2, G2 = 4, 96
    S2 = G2 * G2
    qkv2 = hip.H2(torch.randn(2, B2 * S2, 3 * D, device="cuda").half())
    out2 = hip.H2.empty(B2 * S2, D)
    rg2 = hip.H2((torch.randn(2, 2 * G2 - 1, hd, device="cuda") * 0.1).half())
    run("sam global 96", lambda: hip.attention(qkv2, out2, B2, S2, H, hd, mode=1, grid=G2, rel_h=rg2, rel_w=rg2, split_qk=split[0], split_pv=split[1], head_major=hm),
        4.0 * B2 * H * S2 * S2 * hd)
