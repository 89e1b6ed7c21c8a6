set -x
# L2 (TCC) request counters of every gemm_nt_kernel launch of the default bench command, operands in planes vs in the 128-byte-row images
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_l2c
rm -rf $O; mkdir -p $O
for MODE in planes images; do
  # round 3 also switched the WEIGHT images off here (CVLM_GEMM_WIL=0; result: profiles/r03_l2_requests.log); that switch is gone
  # since round 4 (the images are always used), so "planes" now means planar ACTIVATIONS only
  if [ $MODE = planes ]; then export CVLM_GEMM_AIL=0; else export CVLM_GEMM_AIL=1; fi
  rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$MODE -- python3 bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-roofline --no-power > $O/run_$MODE.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for mode in ("planes", "images"):
    fs = sorted(glob.glob(f"gpurun_out/pmc_l2c/{mode}/**/*_counter_collection.csv", recursive=True))
    tot = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(fs[-1])):
        if "gemm_nt_kernel<3" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_REQ_sum": n += 1
    print(f"{mode:7s}: {n} gemm_nt_kernel<3,...> launches; per launch: TCC_REQ {tot['TCC_REQ_sum']/n/1e6:8.2f} M  READ {tot['TCC_READ_sum']/n/1e6:8.2f} M  HIT {tot['TCC_HIT_sum']/n/1e6:8.2f} M  MISS {tot['TCC_MISS_sum']/n/1e6:8.2f} M")
PY
rm -rf $O/planes $O/images
