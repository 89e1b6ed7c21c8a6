set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_b8t
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/stats -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-power > $O/bench.log 2>&1
python tools/step_timeline.py $O/stats 12 > $O/b8_step_timeline.log 2>&1
rm -rf $O/stats
