"""kernels between the end of a step's evaluation tail and the first encoder kernel of the next step: python trace_window.py <dir>"""
import csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", ""))[:70]
fin = [i for i, r in enumerate(rows) if "wfm_finalize" in r[2]]
# ends of steps: a wfm_finalize followed (within the next 40 kernels) by a resample kernel
ends = [i for i in fin if any("resample" in rows[j][2] for j in range(i + 1, min(len(rows), i + 40)))]
for i in ends[-2:]:
    j = i + 1
    while j < len(rows) and "patchify" not in rows[j][2]:
        j += 1
    t0, prev = rows[i][1], rows[i][1]
    print(f"--- after step end (kernel {i}) to first patchify (kernel {j}): {(rows[j][0] - t0) / 1e3:.1f} us")
    for s, e, n, q in rows[i + 1:j + 1]:
        print(f"+{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:8.1f}  q{q}  {short(n)}")
        prev = max(prev, e)
# biggest idle gaps (union over queues) in the last 40 % of the trace
part = rows[int(len(rows) * 0.6):]
prev_end, prev_name = part[0][1], part[0][2]
gaps = []
for s, e, n, q in part[1:]:
    if s > prev_end:
        gaps.append((s - prev_end, short(prev_name), short(n), s))
    if e > prev_end:
        prev_end, prev_name = e, n
gaps.sort(reverse=True)
print("--- biggest gaps in the last 40 % of the trace")
for g, a, b, s in gaps[:25]:
    print(f"{g / 1e3:10.1f} us at +{(s - part[0][0]) / 1e6:9.3f} ms   {a}  ->  {b}")
pat = [r[0] for r in part if "reinterpret_transpose" in r[2]]
print("step periods (ms) by reinterpret_transpose:", [round((b - a) / 1e6, 2) for a, b in zip(pat, pat[1:])])
