#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries committed under profiles/.

  python tools/summarize_profiles.py stats  <dir-with-*_kernel_stats.csv>  profiles/<name>.csv
  python tools/summarize_profiles.py traffic <fetch-dir> <write-dir> profiles/<name>.json [kernel-substring [tail-fraction]]

`traffic` averages the FETCH_SIZE and WRITE_SIZE counters (collected in separate --pmc passes, KB units) over the
launches of the dominant kernel; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(128-B requests are tallied at 64 B)."""
import csv
import glob
import json
import os
import shutil
import sys


def _one(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[-1]


def _avg_counter(d, counter, needle, tail_frac=1.0):
    """average of `counter` over the launches whose kernel name holds `needle`; tail_frac < 1: only the LAST share of them in
    dispatch order (the steady-state steps: weight packing, text bank and constant GEMMs of the set-up come first)"""
    vals = []
    with open(_one(d, "*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and needle in row["Kernel_Name"]:
                vals.append((int(row.get("Dispatch_Id") or len(vals)), float(row["Counter_Value"])))
    vals.sort()
    vals = [v for _, v in vals][int(len(vals) * (1.0 - tail_frac)):]
    return sum(vals) / max(len(vals), 1), len(vals)


def main():
    if sys.argv[1] == "stats":
        shutil.copyfile(_one(sys.argv[2], "*_kernel_stats.csv"), sys.argv[3])
        print("wrote", sys.argv[3])
    elif sys.argv[1] == "traffic":
        needle = sys.argv[5] if len(sys.argv) > 5 else "gemm_nt_kernel<3"
        tail = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
        fetch_kb, n = _avg_counter(sys.argv[2], "FETCH_SIZE", needle, tail)
        write_kb, m = _avg_counter(sys.argv[3], "WRITE_SIZE", needle, tail)
        out = {"kernel": needle + "...>", "launches_profiled": n, "fetch_bytes_per_launch_corrected_x2": fetch_kb * 1024 * 2,
               "write_bytes_per_launch": write_kb * 1024, "traffic_bytes_per_launch": fetch_kb * 2048 + write_kb * 1024,
               "launch_window": "all launches of the run" if tail >= 1.0 else f"the last {tail:.0%} of the launches in dispatch order (steady-state steps only)",
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the bench command; "
                         "KB->bytes x1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
        assert n == m and n > 0, (n, m)
        with open(sys.argv[4], "w") as f:
            json.dump(out, f, indent=1)
        print(out)


if __name__ == "__main__":
    main()
