set -x
# kernel boundaries of the one-stream step under rocprofv3 --kernel-trace, per precision: where does the GPU idle between dependent launches?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gaps_r6
rm -rf $O; mkdir -p $O
for P in ${PRECS:-exact mx}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t_$P -- python3 bench.py --precision $P --steps 3 --warmup 2 --no-overlap --no-cpu-baseline --no-exact-leg --no-power --no-roofline > $O/bench_$P.log 2>&1
  python tools/kernel_gaps.py $O/t_$P > $O/gaps_$P.log 2>&1
  rm -rf $O/t_$P
  head -14 $O/gaps_$P.log | cut -c1-200
done
