#!/usr/bin/env python3
"""Register / scratch / LDS report per kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

    python tools/kernel_resources.py camouflaged-vlm_amd/csrc/gemm.hip [substring-filter] [-DCVLM_PROBES ...]

A VGPR spill inside a main loop costs scratch traffic on every trip; this is the check after every kernel edit."""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.basename(src), "-o", "/dev/null"] + extra
out = subprocess.run(cmd, cwd=os.path.dirname(os.path.abspath(src)), capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|"
                  r"LDS Size \[bytes/block\]):\s*(\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = {"name": m.group(2)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1).split(" [")[0]] = m.group(2)
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
except OSError:
    dem = [r["name"] for r in rows]
print("%5s %5s %7s %6s %6s %4s  %s" % ("VGPR", "AGPR", "scratch", "vspill", "sspill", "occ", "kernel"))
for r, d in zip(rows, dem):
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"\((?:\(anonymous namespace\)::)?\w*Params\)|\(.*\)$", "", d)
    if flt and not all(f in d for f in flt):
        continue
    print("%5s %5s %7s %6s %6s %4s  %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                          r.get("Occupancy"), d[:150]))
