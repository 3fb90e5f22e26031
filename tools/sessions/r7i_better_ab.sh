#!/bin/bash
# Session r7i: SpeedBetter fused candidate loads (base) vs -DZB_FUSE=0 (zbf0), C5, more repetitions than r7h (whose pipelined runs had a cold context:
# warm-up 2 with three contexts); one context and three contexts.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r7i}
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify"
for rep in 1 2 3; do
for tag in ${TAGS:-zbf0 base}; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    for mode in "--no-pipeline --steps 10 --warmup 3" "--steps 12 --warmup 6"; do
    env $E timeout 300 python bench.py $B $mode 2>$OUT/$tag.err | tail -1 > $OUT/$tag.json
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag $mode |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
    done
done
done
