set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/c -- python3 tools/bench_attn.py > $O/run.log 2>&1
python - <<'PY'
import csv, glob, collections
fs = sorted(glob.glob("gpurun_out/pmc_attn/c/**/*_counter_collection.csv", recursive=True))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fs[-1])):
    k = r["Kernel_Name"]
    key = k.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:40]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_REQ_sum": n[key] += 1
for key, d in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:8]:
    c = max(n[key], 1)
    print(f"{key:42s} launches {c:4d}  per launch: REQ {d.get('TCC_REQ_sum',0)/c:12.0f} READ {d.get('TCC_READ_sum',0)/c:12.0f} WRITE {d.get('TCC_WRITE_sum',0)/c:12.0f} HIT {d.get('TCC_HIT_sum',0)/c:12.0f} MISS {d.get('TCC_MISS_sum',0)/c:12.0f}")
PY
rm -rf $O/c
