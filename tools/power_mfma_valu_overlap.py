"""tools/micro/mfma_valu_overlap.hip with the socket power and shader clock of each configuration (amdgpu hwmon, 20 samples / s):
do the matrix pipe and the vector ALU of a SIMD overlap, and what is the overlap worth at the power cap?
    hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/bin/mfma_valu_overlap
    python tools/power_mfma_valu_overlap.py [seconds per configuration]"""
import glob, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hw = [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(d, "power1_input"))]
samples, stop = [], False
def read(path):
    try:
        return float(open(path).read())
    except Exception:
        return float("nan")
def sampler():
    while not stop:
        # the busiest card is ours (one GPU per box)
        rows = [(read(os.path.join(d, "power1_input")) / 1e6, read(os.path.join(d, "freq1_input")) / 1e6) for d in hw]
        if rows:
            samples.append((time.time(),) + max(rows))
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
out = subprocess.run([os.path.join(ROOT, "tools", "micro", "bin", "mfma_valu_overlap"), sys.argv[1] if len(sys.argv) > 1 else "2.0"], capture_output=True, text=True).stdout
stop = True; th.join()
for line in out.splitlines():
    m = re.search(r"t0 ([0-9.]+) t1 ([0-9.]+)", line)
    xs = [(p, f) for t, p, f in samples if m and float(m.group(1)) + 0.5 <= t <= float(m.group(2)) - 0.1]
    tail = "%.0f W, %.0f MHz (%d samples)" % (sum(p for p, _ in xs) / len(xs), sum(f for _, f in xs) / len(xs), len(xs)) if xs else "no samples"
    ns = float(re.search(r"= +([0-9.]+) ns per trip", line).group(1))
    mhz = sum(f for _, f in xs) / len(xs) if xs else float("nan")
    print(line.split(" | t0")[0], "|", tail, "| %.0f shader cycles per trip" % (ns * mhz * 1e-3), flush=True)
