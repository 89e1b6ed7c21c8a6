"""Is the split-3 GEMM power-limited?  Runs one encoder GEMM shape in a loop for a few seconds per variant and samples
rocm-smi (socket power, sclk) from a child process meanwhile."""
import os, sys, time, subprocess, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip
hip.load()
ws = hip.new_gemm_workspace("cuda")
M, N, K = 32768, 5120, 1280
a = hip.H2(torch.randn(2, M, K, device="cuda").half())
w = hip.H2((torch.randn(2, N, K, device="cuda") * 0.05).half())
oh = hip.H2.empty(M, N)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), out))
        except Exception as e:
            samples.append((time.time(), "ERR %r" % e))
        time.sleep(0.3)
def summarize(tag, t0, t1):
    import re
    pw, sc = [], []
    for t, out in samples:
        if t0 + 1.0 <= t <= t1:
            for m in re.finditer(r"Power \(W\): ([0-9.]+)", out): pw.append(float(m.group(1)))
            for m in re.finditer(r"sclk clock level: \d+: \((\d+)Mhz\)", out): sc.append(int(m.group(1)))
    print(tag, "power W:", pw[:12], "sclk MHz:", sc[:12], flush=True)
th = threading.Thread(target=sampler); th.start()
time.sleep(2.0)
t_idle0 = time.time(); time.sleep(2.0); summarize("idle", t_idle0 - 1.0, time.time())
for tag, env in (("persist=0", "0"), ("persist=1", "1"), ("split1 (hi only)", None)):
    if env is not None: os.environ["CVLM_GEMM_PERSIST"] = env
    kw = dict(split=1) if env is None else {}
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 5.0:
        for _ in range(50): hip.gemm(a, w, M, N, K, out_h2=oh, workspace=ws, **kw)
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    t1 = time.time()
    print(tag, "avg us/launch", e0.elapsed_time(e1) * 1e3 / n, flush=True)
    summarize(tag, t0, t1)
stop = True; th.join()
if samples: print(samples[len(samples)//2][1][:1500])
