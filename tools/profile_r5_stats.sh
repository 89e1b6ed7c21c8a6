set -x
# kernel trace + stats of the default bench command in the mx precision (one stream), summarised per kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r5
rm -rf $O; mkdir -p $O
P=${PRECISION:-mx}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --precision $P --steps 4 --warmup 2 --no-overlap --no-cpu-baseline --no-power > $O/bench_stats.log 2>&1
python tools/summarize_profiles.py stats $O/stats $O/r05_bench_${P}_b8_kernel_stats.csv
python tools/kernel_gaps.py $O/stats > $O/r05_kernel_gaps.log 2>&1
rm -rf $O/stats
head -40 $O/r05_bench_${P}_b8_kernel_stats.csv
