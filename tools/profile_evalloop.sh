set -x
# kernel trace of the evaluation loop (bench.py --surface evalloop): which kernels the N1 / N2 sections spend their time in
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-8}
O=gpurun_out/prof_evalloop_b$B
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --surface evalloop --batch $B --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-power > $O/bench_stats.log 2>&1
python tools/summarize_profiles.py stats $O/stats $O/r04_evalloop_b${B}_kernel_stats.csv
rm -rf $O/stats
