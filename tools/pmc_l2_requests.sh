set -x
# L2 (TCC) request counters of the big GEMM shapes with the operands staged from planes vs from the 128-byte-row images
# (weights: eighth field of the tools/ab_gemm.py variant string, 1 / 0 = with / without the weight image; the probe passes the weight image only -- activations of tools/ab_gemm.py are planar).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_l2
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -o "TCC_[A-Z_0-9]*REQ[A-Za-z_0-9]*\|TCP_TCC_READ_REQ[A-Za-z_0-9]*\|TCC_HIT[A-Za-z_0-9]*\|TCC_MISS[A-Za-z_0-9]*" $O/counters.txt | sort -u | head -40 > $O/counter_names.txt
for WL in 0 1; do
  rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/w$WL -- python3 tools/ab_gemm.py 0:1:0:1:4:1:1:$WL > $O/run_w$WL.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for wl in (0, 1):
    fs = sorted(glob.glob(f"gpurun_out/pmc_l2/w{wl}/**/*_counter_collection.csv", recursive=True))
    if not fs:
        print("no counter file for", wl); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"]
        if "gemm_nt_kernel" not in k: continue
        key = (k.split("(")[0][-60:], r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "TCC_REQ_sum": n[key] += 1
    print(f"== CVLM_GEMM_WIL={wl}")
    for key, d in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:6]:
        c = max(n[key], 1)
        print(f"{key[0]:62s} grid {key[1]:>8s} launches {c:4d}  per launch: TCC_REQ {d.get('TCC_REQ_sum',0)/c:14.0f}  READ {d.get('TCC_READ_sum',0)/c:14.0f}  HIT {d.get('TCC_HIT_sum',0)/c:14.0f}  MISS {d.get('TCC_MISS_sum',0)/c:12.0f}")
PY
rm -rf $O/w0 $O/w1
