#!/bin/bash
# Builds of the ViT-H attention kernels with one operand's lo plane dropped from its product (common.h CVLM_ATTN_TERMS), next to the
# product library: lib_terms/<name>/libcvlm_hip.so (the other objects are the product build's).  Run on the GPU box with
#   tools/ab_attn_terms.sh run   ->  per build: the 16-image digest gate of tests/test_cascade_mx_gpu.py and one bench.py line
set -e
cd "$(dirname "$0")/../camouflaged-vlm_amd/csrc"
# CVLM_ATTN_TERMS (common.h): bit 0 P, 1 Q, 2 K, 3 V = that operand's lo plane takes part; the builds below drop the named ones
VARIANTS=("p:-DCVLM_ATTN_TERMS=14" "q:-DCVLM_ATTN_TERMS=13" "k:-DCVLM_ATTN_TERMS=11" "v:-DCVLM_ATTN_TERMS=7" "pq:-DCVLM_ATTN_TERMS=12"
          "pk:-DCVLM_ATTN_TERMS=10" "pqk:-DCVLM_ATTN_TERMS=8" "none:-DCVLM_ATTN_TERMS=15")
if [ "$1" != "run" ]; then
    make -j4 >/dev/null
    for v in "${VARIANTS[@]}"; do
        n=${v%%:*}; f=${v#*:}; d=../lib_terms/$n; mkdir -p $d
        for src in attention_g64pp attention_win2; do
            /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $f -c $src.hip -o $d/$src.o &
        done
        wait
        objs=""
        for o in gemm gemm_il gemm_mx rowops attention preprocess evaltail; do objs="$objs ../lib/$o.o"; done
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libcvlm_hip.so $objs $d/attention_g64pp.o $d/attention_win2.o
        echo "built $d"
    done
    exit 0
fi
cd ../..
set +e
OUT=${OUT:-gpurun_out/attn_terms}; mkdir -p $OUT
for n in ${NAMES:-none p q k v pq pk pqk}; do
    export CVLM_PROBE_LIB=$PWD/camouflaged-vlm_amd/lib_terms/$n/libcvlm_hip.so      # `none` = all four lo planes in (the round's earlier build)
    echo "== $n" | tee -a $OUT/summary.log
    python -m pytest tests/test_cascade_mx_gpu.py -q -s -k "all_16" 2>&1 | grep -E "mx, demo|mx, 1536|passed|failed|Error|assert" | tee -a $OUT/summary.log
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_$n.json 2>$OUT/bench_$n.err
    python - $OUT/bench_$n.json <<'PY' | tee -a $OUT/summary.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("   %.2f img/s  %.2f ms/step  parity %s" % (d["value"], d["ms_per_step"], {k: d.get("parity", {}).get(k) for k in ("max_abs_mask_err", "max_abs_class_logit_err", "mask_iou")}))
print("   attention:", json.dumps(r.get("secondary"))[:900])
PY
done
