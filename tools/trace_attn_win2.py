"""Where a (window, head) pair's time goes in the producer / consumer window-attention kernel (attention_win2.hip), B = 8 cascade
shape, on a probe build:
  make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES LIBDIR=../lib_probe
  CVLM_PROBE_LIB=camouflaged-vlm_amd/lib_probe/libcvlm_hip.so python tools/trace_attn_win2.py
Wave 0 of every workgroup stamps the wall clock (100 MHz) around the three stages of each pair; the sums per workgroup come back."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
lib = hip.load()
B, H, hd, G = 8, 16, 80, 64
SPLIT = int(os.environ.get("SPLIT", "2"))
D, S = H * hd, G * G
qkv = hip.H2(torch.randn(2, B * S, 3 * D, device="cuda").half())
out = hip.H2.empty(B * S, D)
rw = hip.H2((torch.randn(2, 27, hd, device="cuda") * 0.1).half())
pad = hip.H2((torch.randn(2, 3 * D, device="cuda") * 0.1).half())
fn = lambda: hip.attention(qkv, out, B, S, H, hd, mode=2, grid=G, window=14, pad=pad, rel_h=rw, rel_w=rw, split_qk=SPLIT, split_pv=SPLIT, head_major=True, scale=1.0)
nwg = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f"launch without stamps: {1e3 * e0.elapsed_time(e1):.1f} us")
assert lib.cvlm_debug_set_attn_win2_trace(C.c_void_p(buf.data_ptr())) == 0
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
assert lib.cvlm_debug_set_attn_win2_trace(None) == 0
print(f"launch with stamps:    {1e3 * e0.elapsed_time(e1):.1f} us")
raw = buf.cpu().numpy().reshape(nwg, 8, 8)
simd = (raw[:, :, 7] >> 4) & 3
print("SIMD of waves 0..7 (workgroups 0, 1, 100):", simd[0].tolist(), simd[1].tolist(), simd[100].tolist())
for w in range(7):
    tw = raw[:, w].astype(np.float64)
    print(f"  wave {w}: tiles {(tw[:, 1] / tw[:, 3]).mean() / 100:6.2f} us per pair, of which at the barriers {(tw[:, 6] / tw[:, 3]).mean() / 100:5.2f}; "
          f"U {(tw[:, 0] / tw[:, 3]).mean() / 100:5.2f}, output {(tw[:, 2] / tw[:, 3]).mean() / 100:5.2f}")
t = raw[:, 0].astype(np.float64)
pairs = t[:, 3]
us = lambda x: x / 100.0
print(f"{nwg} workgroups, {int(pairs.sum())} pairs ({int(pairs.min())}-{int(pairs.max())} per workgroup); first stamp to last stamp: "
      f"mean {us(t[:, 5] - t[:, 4]).mean():.1f} us, span over the chip {us(t[:, 5].max() - t[:, 4].min()):.1f} us")
for name, col in (("U = Q.R^T + scatter + augmented fragments", 0), ("seven key tiles", 1), ("  of which at the step barriers", 6), ("output", 2)):
    v = us(t[:, col] / pairs)
    print(f"  per pair: {name:44s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
tot = us((t[:, 0] + t[:, 1] + t[:, 2]) / pairs)
print(f"  per pair: total {tot.mean():.2f} us; x {pairs.max():.0f} pairs = {tot.mean() * pairs.max():.1f} us")
