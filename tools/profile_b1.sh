set -x
# kernel trace of the drop-in surface at batch 1 (the reference's own call pattern): per-kernel stats, idle gaps, serial tail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r04}
O=gpurun_out/prof_b1
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --surface dropin --batch 1 --steps 16 --warmup 2 --no-cpu-baseline --no-roofline --no-power > $O/bench_stats.log 2>&1
python tools/summarize_profiles.py stats $O/stats $O/${R}_dropin_b1_kernel_stats.csv
python tools/kernel_gaps.py $O/stats tail:0.25 > $O/${R}_dropin_b1_kernel_gaps.log 2>&1
python tools/step_timeline.py $O/stats 16 > $O/${R}_dropin_b1_step_timeline.log 2>&1
python tools/b1_tail.py $O/stats --list > $O/${R}_dropin_b1_tail.log 2>&1
rm -rf $O/stats
