"""Phase times of the ping-pong global-attention kernel (B = 8 cascade shape).  Usage: [SPLIT=3|2] python tools/trace_attn_g64.py"""
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
rg = hip.H2((torch.randn(2, 2 * G - 1, hd, device="cuda") * 0.1).half())
fn = lambda: hip.attention(qkv, out, B, S, H, hd, mode=1, grid=G, rel_h=rg, rel_w=rg, split_qk=SPLIT, split_pv=SPLIT, head_major=True)
nwave = 16 * H * B * 8
buf = torch.zeros(nwave * 8, dtype=torch.int64, device="cuda")
for _ in range(2): fn()
torch.cuda.synchronize()
assert lib.cvlm_debug_set_attn_g64_trace(C.c_void_p(buf.data_ptr())) == 0
fn(); torch.cuda.synchronize()
assert lib.cvlm_debug_set_attn_g64_trace(None) == 0
raw = buf.cpu().numpy().reshape(-1, 8, 8)
if raw[:, :, 6].max() > 0:
    print(f"shader clock over the kernel: {(raw[:, :, 6] / (raw[:, :, 5] / 100.0)).mean() / 1e3:.3f} GHz (s_memtime ticks / wall)")
print(f"split {SPLIT}")
t = raw / 100.0          # [wg][wave][field] in us
for grp, lab in ((slice(0, 4), "group A (waves 0-3)"), (slice(4, 8), "group B (waves 4-7)")):
    v = t[:, grp, :].reshape(-1, 8)
    print(f"{lab}: prologue {v[:,0].mean():6.1f}  X {v[:,1].mean():6.1f}  X wait+barrier {v[:,2].mean():6.1f}  "
          f"Y {v[:,3].mean():6.1f}  Y wait+barrier {v[:,4].mean():6.1f}  total {v[:,5].mean():6.1f} us  (128 tiles)")
