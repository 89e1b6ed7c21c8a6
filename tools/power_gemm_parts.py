"""Where the GEMM's power goes: probe variants of the 256^2 kernel (probe build) under rocm-smi sampling.
CVLM_PROBE_LIB=camouflaged-vlm_amd/lib_probe/libcvlm_hip.so python tools/power_gemm_parts.py"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
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
        time.sleep(0.25)
def window(t0, t1):
    xs = [(p, s) for t, p, s in samples if t0 + 1.0 <= t <= t1 - 0.2]
    if not xs: return "no samples"
    return "power %.0f W, sclk %.0f MHz (%d samples)" % (sum(p for p, _ in xs) / len(xs), sum(s for _, s in xs) / len(xs), len(xs))
th = threading.Thread(target=sampler); th.start()
sys.path.insert(0, ROOT)
import torch
from camouflaged_vlm_amd import hip
hip.load()
ws = hip.new_gemm_workspace("cuda")
M, N, K = 32768, 5120, 1280
a = hip.H2(torch.randn(2, M, K, device="cuda").half()); w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half())
oh = hip.H2.empty(M, N)
for tag, var in (("product kernel (non-persistent)", "7"), ("no DMA in the steady state (MFMA + LDS reads + epilogue)", "17"),
                 ("DMA only (no LDS reads, no MFMA)", "27"), ("no epilogue stores", "37"), ("main loop only", "67"),
                 ("DMA only, 128^2 tile, 64-byte rows (BK = 32)", "21"), ("DMA only, 128^2 tile, 128-byte rows (BK = 64)", "24")):
    os.environ["CVLM_GEMM_VARIANT"] = var
    for _ in range(3): hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50): hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws)
        n += 50; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    print("%-60s %7.1f us/launch | %s" % (tag, e0.elapsed_time(e1) * 1e3 / n, window(t0, time.time())), flush=True)
stop = True; th.join()
