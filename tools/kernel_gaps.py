#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 --kernel-trace run (one stream): where the GPU waits for the host.
  python tools/kernel_gaps.py <dir-with-*_kernel_trace.csv> [n_last_kernels_of_interest]"""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", ""))[:60]
# steady-state steps = from the middle to the 95th percentile of the GEMM launches (the head of a trace holds weight uploads and the
# on-device weight packing -- thousands of torch kernels since round 5 --, the tail the parity check and CPU work);
# "tail:<frac>" as second argument: the last <frac> of the kernels instead (short runs whose head is mostly uploads)
n = len(rows)
gi = [i for i, r in enumerate(rows) if "gemm_nt_kernel" in r[2]]
part = rows[gi[len(gi) // 2]: gi[len(gi) * 95 // 100]] if len(gi) >= 100 else rows[n // 3: n * 2 // 3]
if len(sys.argv) > 2 and sys.argv[2].startswith("tail:"):
    part = rows[int(n * (1.0 - float(sys.argv[2][5:]))):]
    # per-kernel busy time, union over overlapping streams
    ivs = sorted((s, e) for s, e, _ in part)
    cover, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
    for s, e in ivs[1:]:
        if s > cur_e:
            cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    print(f"tail window: {len(part)} kernels, wall {(ivs[-1][1] - ivs[0][0]) / 1e6:.2f} ms, GPU busy (union of kernels) {cover / 1e6:.2f} ms, "
          f"sum of kernel durations {sum(e - s for s, e in ivs) / 1e6:.2f} ms")
busy = sum(e - s for s, e, _ in part)
wall = part[-1][1] - part[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
big = 0
for (s0, e0, n0), (s1, e1, n1) in zip(part, part[1:]):
    g = s1 - e0
    if g > 0:
        k = short(n0) + "  ->  " + short(n1)
        gaps[k][0] += 1
        gaps[k][1] += g
        big += g
print(f"{len(part)} kernels: wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {big / 1e6:.2f} ms ({100.0 * big / wall:.1f} %), "
      f"mean gap {big / max(1, len(part) - 1) / 1e3:.2f} us")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e3:9.1f} us in {c:5d} gaps ({t / c / 1e3:6.2f} us each)  {k}")
