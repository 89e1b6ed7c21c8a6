"""Power roofline of the matrix pipe only (tools/micro/mfma_power.hip): socket power and shader clock
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
    out = subprocess.run([binp, "4"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.search(r"t0 ([0-9.]+) t1 ([0-9.]+)", line)
        print(line.split(" | t0")[0], "|", line.split("|")[-1].strip(), "|", window(float(m.group(1)), float(m.group(2))) if m else "", flush=True)
stop = True; th.join()
