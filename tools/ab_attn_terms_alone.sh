#!/bin/bash
# One attention operand's lo plane dropped at a time on top of the split-3 engine (--precision exact): the "alone" column of the table
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/attn_terms}; mkdir -p $OUT
for n in ${NAMES:-none p q pq}; do
    export CVLM_PROBE_LIB=$PWD/camouflaged-vlm_amd/lib_terms/$n/libcvlm_hip.so
    python bench.py --precision exact --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_exact_$n.json 2>$OUT/bench_exact_$n.err
    python - $OUT/bench_exact_$n.json $n <<'PY' | tee -a $OUT/summary_alone.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d.get("parity", {})
print("exact + %-4s %.2f img/s  mask %.3e  class logits %.3e  IoU %.6f  pred_equal %s" % (sys.argv[2], d["value"], p.get("max_abs_mask_err"), p.get("max_abs_class_logit_err"), p.get("mask_iou"), p.get("pred_equal")))
PY
done
