"""Power roofline of the matrix pipe (tools/micro/mfma_power.hip) and of the split-3 GEMM: socket power and shader clock
(rocm-smi, sampled every 0.3 s) while each loop runs for a few seconds."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, stop = [], False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = [float(m.group(1)) for m in re.finditer(r"Power \(W\): ([0-9.]+)", out)]
            sc = [int(m.group(1)) for m in re.finditer(r"sclk clock level: \d+: \((\d+)Mhz\)", out)]
            if pw and sc: samples.append((time.time(), pw[0], sc[0]))
        except Exception:
            pass
        time.sleep(0.3)
def window(t0, t1):
    xs = [(p, s) for t, p, s in samples if t0 + 1.0 <= t <= t1 - 0.2]
    if not xs: return "no samples"
    return "power %.0f W (min %.0f, max %.0f), sclk %.0f MHz (min %d, max %d), %d samples" % (
        sum(p for p, _ in xs) / len(xs), min(p for p, _ in xs), max(p for p, _ in xs),
        sum(s for _, s in xs) / len(xs), min(s for _, s in xs), max(s for _, s in xs), len(xs))
th = threading.Thread(target=sampler); th.start()
time.sleep(1.5)
t0 = time.time(); time.sleep(2.5); print("idle:", window(t0 - 1.0, time.time() + 0.2), flush=True)
binp = os.path.join(ROOT, "tools", "micro", "bin", "mfma_power")
if os.path.exists(binp):
    order_study = os.environ.get("MFMA_ORDER") in ("1", "2", "3", "4")  # operand reuse between consecutive MFMAs only (mfma_power.hip k_order / k_gemm)
    out = subprocess.run([binp, "4"] + ([os.environ["MFMA_ORDER"]] if order_study else []), capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.search(r"t0 ([0-9.]+) t1 ([0-9.]+)", line)
        print(line.split(" | t0")[0], "|", line.split("|")[-1].strip(), "|", window(float(m.group(1)), float(m.group(2))) if m else "", flush=True)
if os.path.exists(binp) and order_study:
    stop = True; th.join(); sys.exit(0)
# the product GEMM at the encoder's lin1 shape, in this process
sys.path.insert(0, ROOT)
import torch
from camouflaged_vlm_amd import hip
hip.load()
ws = hip.new_gemm_workspace("cuda")
M, N, K = 32768, 5120, 1280
a = hip.H2(torch.randn(2, M, K, device="cuda").half()); w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half())
oh = hip.H2.empty(M, N)
for tag, kw in (("gemm 32768x5120x1280 split 3 (exact)", {}), ("gemm 32768x5120x1280 split 1 (hi planes only)", dict(split=1))):
    for _ in range(3): hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws, **kw)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50): hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws, **kw)
        n += 50; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    alg = 2.0 * M * N * K / us / 1e6
    print("%s | %.1f us/launch = %.0f TFLOP/s algorithmic, %.0f issued | %s" % (tag, us, alg, alg * (1 if kw else 3), window(t0, time.time())), flush=True)
# the two SAM attention kernels (B = 8)
B, H, hd, G = 8, 16, 80, 64
D, S = H * hd, G * G
qkv = hip.H2(torch.randn(2, B * S, 3 * D, device="cuda").half()); out = hip.H2.empty(B * S, D)
rg = hip.H2((torch.randn(2, 2 * G - 1, hd, device="cuda") * 0.1).half()); rw = hip.H2((torch.randn(2, 27, hd, device="cuda") * 0.1).half())
pad = hip.H2((torch.randn(2, 3 * D, device="cuda") * 0.1).half())
for tag, fn, fl in (("global attention 64x64 map", lambda: hip.attention(qkv, out, B, S, H, hd, mode=1, grid=G, rel_h=rg, rel_w=rg, split_qk=3, split_pv=3, head_major=True), 4.0 * B * H * S * S * hd),
                    ("window attention 14x14", lambda: hip.attention(qkv, out, B, S, H, hd, mode=2, grid=G, window=14, pad=pad, rel_h=rw, rel_w=rw, split_qk=3, split_pv=3, head_major=True), 4.0 * B * H * 25 * 196 * 196 * hd)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(20): fn()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("%s | %.1f us/launch = %.0f TFLOP/s algorithmic | %s" % (tag, us, fl / us / 1e6, window(t0, time.time())), flush=True)
stop = True; th.join()
