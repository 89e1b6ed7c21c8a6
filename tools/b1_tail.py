#!/usr/bin/env python3
"""The serial tail of one drop-in step at batch 1 (rocprofv3 --kernel-trace): every kernel from the last global-attention launch of
the encoder to the stage-2 clip_head, with start offset, duration and the idle gap in front of it; then the totals per region
(rest of the encoder | decoder | stage 2).   python tools/b1_tail.py <dir-with-*_kernel_trace.csv> [--list]"""
import csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id") or "0"))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", ""))[:60]
heads = [i for i, r in enumerate(rows) if "clip_head_kernel" in r[2]]
i_end = heads[-3]                                         # stage-2 head of the last complete step but one
i_prev = heads[-5]
step = rows[i_prev + 1:i_end + 1]
g4 = [i for i, r in enumerate(step) if "attn_g64" in r[2]][-1]
tail = step[g4:]
t0 = tail[0][0]
dec0 = next(i for i, r in enumerate(tail) if "dense_pe" in r[2] or "small_attn" in r[2] or ("split_kernel" in r[2] and i > 12))
# decoder region: from the first kernel after the neck's last layernorm; stage 2 from the sigmoid / resize in front of the CLIP forward
ln = [i for i, r in enumerate(tail[:dec0 + 1]) if "layernorm" in r[2]]
dec0 = ln[-1] + 1 if ln else dec0
st2 = next(i for i, r in enumerate(tail) if i > dec0 and ("sigmoid" in r[2] or "upsample" in r[2]))
regions = [("encoder after the last global attention", 0, dec0), ("mask decoder + postprocess", dec0, st2), ("stage 2 (glue + CLIP forward + head)", st2, len(tail))]
prev = tail[0][0]
if "--list" in sys.argv:
    for i, (s, e, n, q) in enumerate(tail):
        tag = [r[0] for r in regions if r[1] == i]
        if tag:
            print(f"---- {tag[0]}")
        print(f"+{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {short(n)}")
        prev = max(prev, e)
print(f"step wall {(step[-1][1] - rows[i_prev][1]) / 1e6:.3f} ms, {len(step)} kernels")
for name, a, b in regions:
    part = tail[a:b]
    busy = sum(e - s for s, e, _, _ in part)
    wall = part[-1][1] - (tail[a - 1][1] if a else part[0][0])
    print(f"{name:42s} {len(part):4d} kernels  wall {wall / 1e3:8.1f} us  kernel time {busy / 1e3:8.1f} us  idle {(wall - busy) / 1e3:8.1f} us  ({(wall - busy) / 1e3 / max(1, len(part)):.1f} us per kernel)")
