set -x
# Round-6 evidence run (one MI355X box): kernel trace + stats of the default bench command (precision mx, fused CLIP forward, one stream),
# HBM traffic and SQ counters of the dominant kernels in separate --pmc passes, the secondary bench lines, per-shape times.
# Everything lands in gpurun_out/prof_r6/; the summaries are copied into profiles/r06_* by hand afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 4 --warmup 2 --no-overlap --no-cpu-baseline --no-exact-leg --no-power > $O/bench_stats.log 2>&1
python tools/summarize_profiles.py stats $O/stats $O/r06_bench_mx_b8_kernel_stats.csv
python tools/kernel_gaps.py $O/stats > $O/r06_kernel_gaps.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 4 --warmup 1 --no-overlap --no-cpu-baseline --no-exact-leg --no-roofline --no-power > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 4 --warmup 1 --no-overlap --no-cpu-baseline --no-exact-leg --no-roofline --no-power > $O/bench_write.log 2>&1
python tools/summarize_profiles.py traffic $O/fetch $O/write $O/r06_gemm_traffic.json "gemm_nt_kernel<3" 0.5
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-exact-leg --no-roofline --no-power > $O/bench_sq.log 2>&1
python - <<'PY'
import csv, glob, collections, json
f = sorted(glob.glob("gpurun_out/prof_r6/sq/**/*_counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    k = "gemm 256^2 mx (ring)" if ("gemm_nt_kernel<3, 2, 4, 5" in k and "true, true, true>" in k) \
        else "gemm 256^2 split-3" if "gemm_nt_kernel<3, 2, 4, 5" in k else "gemm 256x128" if "gemm_nt_kernel<3, 4, 2, 3" in k else "gemm 128^2" if "gemm_nt_kernel<3, 2, 2, 2" in k \
        else "gemm small-grid (deep ring, eight waves)" if "gemm_nt_kernel<3, 4, 2, 14" in k \
        else "attn global" if "g64pair" in k else "attn window" if "attn_win14" in k else "attn clip" if "attn_kernel" in k else None
    if k is None: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
out = {}
for k, d in agg.items():
    # SQ_WAVE_CYCLES etc. count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs
    gui = d["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = {"launches": cnt[k], "mfma_busy_frac_of_simd_time": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0) if gui else None,
              "wait_any_frac": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], "wait_inst_any_frac": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
              "active_inst_any_frac": d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], "gui_active_cycles_per_launch": gui / max(cnt[k], 1)}
json.dump(out, open("gpurun_out/prof_r6/r06_sq_counters.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
rm -rf $O/stats $O/fetch $O/write $O/sq
python tools/gemm_shapes.py --batch 8 --reps 2 --pipelined > $O/r06_per_shape_times.log 2>&1
python bench.py --workload encoder --steps 8 --warmup 2 > $O/r06_bench_encoder_b8.json 2> $O/enc.err; tail -c 400 $O/r06_bench_encoder_b8.json
python bench.py --geometry hires1536 --batch 4 --workload encoder --steps 6 --warmup 2 > $O/r06_bench_hires1536_b4.json 2> $O/hires.err; tail -c 400 $O/r06_bench_hires1536_b4.json
python bench.py --surface dropin --batch 1 --steps 32 --no-cpu-baseline > $O/r06_bench_dropin_b1.json 2> $O/dropin_b1.err; tail -c 300 $O/r06_bench_dropin_b1.json
python bench.py --surface dropin --batch 8 --steps 10 --no-cpu-baseline > $O/r06_bench_dropin_b8.json 2> $O/dropin_b8.err; tail -c 300 $O/r06_bench_dropin_b8.json
python tools/gemm_shapes.py --batch 1 --reps 4 > $O/r06_per_shape_times_b1.log 2>&1
python bench.py > $O/r06_bench_default.json 2> $O/default.err; tail -c 600 $O/r06_bench_default.json
for s in 33 22 12; do python tools/bench_attn.py $s; done > $O/r06_bench_attn.log 2>&1
ls -la $O
