"""Per-workgroup timeline of the window-attention kernel (B = 8 cascade shape) on a probe build:
  make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES LIBDIR=../lib_probe
  CVLM_PROBE_LIB=camouflaged-vlm_amd/lib_probe/libcvlm_hip.so python tools/trace_attn_win.py"""
import ctypes as C, os, sys
# the timeline stamps live in the two-workgroups-per-pair kernel (attention_win.hip), which since round 4 runs for every precision
# but exact-with-h2-output (that one is the producer / consumer form, attention_win2.hip): the trace below uses split_pv = 1
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
lib = hip.load()
B, H, hd, G = 8, 16, 80, 64
D, S = H * hd, G * G
qkv = hip.H2(torch.randn(2, B * S, 3 * D, device="cuda").half())
out = hip.H2.empty(B * S, D)
rw = hip.H2((torch.randn(2, 27, hd, device="cuda") * 0.1).half())
pad = hip.H2((torch.randn(2, 3 * D, device="cuda") * 0.1).half())
fn = lambda: hip.attention(qkv, out, B, S, H, hd, mode=2, grid=G, window=14, pad=pad, rel_h=rw, rel_w=rw, split_qk=3, split_pv=1, head_major=True)
nwg = 2 * H * B * 25
buf = torch.zeros(2 * nwg * 8, dtype=torch.int64, device="cuda")
for _ in range(2): fn()
torch.cuda.synchronize()
assert lib.cvlm_debug_set_attn_win_trace(C.c_void_p(buf.data_ptr())) == 0
fn(); torch.cuda.synchronize()
assert lib.cvlm_debug_set_attn_win_trace(None) == 0
t = buf.cpu().numpy()[:nwg * 8].reshape(nwg, 8)
x = buf.cpu().numpy()[nwg * 8:].reshape(nwg, 8)
us = lambda x: x / 100.0
print(f"{nwg} workgroups, span {us(t[:, 3].max() - t[:, 0].min()):.1f} us")
halfbit = (np.arange(nwg) >> 3) & 1    # 1-D grid (round 3): workgroups id and id + 8 are the two query halves of a pair
for half, lab in ((0, "128-query workgroups"), (1, "68-query workgroups")):
    tt = t[halfbit == half]
    for name, v in (("  loads+offsets+dma issue", us(tt[:, 4] - tt[:, 0])), ("  U mfma+scatter+aug", us(tt[:, 5] - tt[:, 4])),
                    ("  wait tiles + scores(0)", us(tt[:, 1] - tt[:, 5])), ("prologue", us(tt[:, 1] - tt[:, 0])),
                    ("7-tile loop", us(tt[:, 2] - tt[:, 1])), ("output", us(tt[:, 3] - tt[:, 2])), ("total", us(tt[:, 3] - tt[:, 0]))):
        print(f"  {lab:22s} {name:26s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
names = ["global loads issued", "offsets computed", "barrier passed", "dma issued (ta)", "loads back", "U + scatter", "aug frags (tb)"]
pts = np.stack([x[:, 0], x[:, 1], x[:, 2], t[:, 4], x[:, 3], x[:, 4], t[:, 5]], 1)
prev = t[:, 0]
for i, nme in enumerate(names):
    d = us(pts[:, i] - prev)
    print(f"    + {nme:22s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    prev = pts[:, i]
# occupancy over time: how many workgroups are inside their tile loop / prologue at a sample of instants
t0, t1 = t[:, 0].min(), t[:, 3].max()
for frac in (0.25, 0.5, 0.75):
    x = t0 + (t1 - t0) * frac
    alive = ((t[:, 0] <= x) & (t[:, 3] > x)).sum()
    loop = ((t[:, 1] <= x) & (t[:, 2] > x)).sum()
    print(f"  at {frac:.2f} of the span: {alive} workgroups resident, {loop} in the tile loop")
