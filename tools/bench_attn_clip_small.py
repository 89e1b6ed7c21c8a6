"""CLIP tower attention (S = 581, 16 heads, head dim 64) at the small batches of the reference's own call pattern: python tools/bench_attn_clip_small.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
hip.load()
for B in (1, 2, 8, 16):
    Hc, hc, Sc = 16, 64, 581
    q2 = hip.H2(torch.randn(2, B * Sc, 3 * Hc * hc, device="cuda").half())
    o2 = hip.H2.empty(B * Sc, Hc * hc)
    fn = lambda: hip.attention(q2, o2, B, Sc, Hc, hc, mode=0)
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us", flush=True)
