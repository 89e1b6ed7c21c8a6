#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace run, as a timeline: where the wall clock of the step goes when two streams
overlap (the drop-in surface at batch 1: CLIP pass 1 under the encoder, then decoder and stage 2 alone).
A step ends with the second clip_head_kernel of a pair (stage 2); the last complete step but one is shown.
  python tools/step_timeline.py <dir-with-*_kernel_trace.csv> [n_gaps]"""
import collections, csv, glob, os, re, sys
d = sys.argv[1]
ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        q = r.get("Stream_Id") or r.get("Queue_Id") or "0"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", ""))[:64]
heads = [i for i, r in enumerate(rows) if "clip_head_kernel" in r[2]]
if len(heads) < 6:
    sys.exit("fewer than three steps in the trace")
i0, i1 = heads[-5] + 1, heads[-3] + 1                      # kernels of the last complete step but one
step = rows[i0:i1]
t0, t1 = rows[heads[-5]][1], step[-1][1]
print(f"step: {len(step)} kernels, wall {1e-6 * (t1 - t0):.3f} ms (end of the previous step's last kernel to the end of this step's last)")
# union busy time and the gaps of the union
ivs = sorted((s, e) for s, e, _, _ in step)
cover, cs, ce, gaps = 0, ivs[0][0], ivs[0][1], [(ivs[0][0] - t0, t0, "previous step's last kernel")]
byend = sorted(step, key=lambda r: r[1])
for s, e in ivs[1:]:
    if s > ce:
        cover += ce - cs
        gaps.append((s - ce, ce, None))
        cs, ce = s, e
    else:
        ce = max(ce, e)
cover += ce - cs
print(f"GPU busy (union over streams) {1e-6 * cover:.3f} ms, idle {1e-6 * (t1 - t0 - cover):.3f} ms in {len(gaps)} gaps; sum of kernel durations {1e-6 * sum(e - s for s, e in ivs):.3f} ms")
streams = collections.OrderedDict()
for s, e, n, q in step:
    streams.setdefault(q, [0, 0, s, e])
    streams[q][0] += 1; streams[q][1] += e - s; streams[q][3] = max(streams[q][3], e)
for q, (c, b, s, e) in streams.items():
    print(f"  stream/queue {q}: {c} kernels, busy {1e-6 * b:.3f} ms, active from +{1e-6 * (s - t0):.3f} to +{1e-6 * (e - t0):.3f} ms")
print(f"largest {ngaps} idle gaps (union):")
for g, at, _ in sorted(gaps, reverse=True)[:ngaps]:
    before = max((r for r in step if r[1] <= at), key=lambda r: r[1], default=None)
    after = min((r for r in step if r[0] >= at + g), key=lambda r: r[0], default=None)
    print(f"  {1e-3 * g:8.1f} us at +{1e-6 * (at - t0):7.3f} ms   {short(before[2]) if before else '(step start)'}  ->  {short(after[2]) if after else '?'}")
# the serial part: everything after the side stream went quiet
main_q = max(streams, key=lambda q: streams[q][1])
side_end = max((v[3] for q, v in streams.items() if q != main_q), default=t0)
print(f"main stream {main_q}; other streams quiet after +{1e-6 * (side_end - t0):.3f} ms")
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in step:
    fam[short(n)][0] += 1; fam[short(n)][1] += e - s
print("kernel time of the step by name:")
for k, (c, b) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {1e-6 * b:7.3f} ms  {c:5d} x {1e-3 * b / c:8.1f} us  {k}")
