#!/bin/bash
# Round 4: the entropy stage at 5 workgroups per CU (sequence chunks of 768 / 512 bring its LDS under 32 KiB) against 4 — after the fix of the
# literal-gather window, which was sized for chunks of 1024 (the fault of r4c).  One context, so that entropy_kernel_ms is the kernel alone.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4n
mkdir -p $OUT
cd $R
B="--config C2 --no-pipeline --no-also --no-cpu-baseline --no-end-to-end --steps 5 --warmup 2"
for tag in base w4c768 w5c768 w5c512 base; do
    E=""; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 200 python bench.py $B > $OUT/${tag}.json 2> $OUT/${tag}.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/${tag}.err").read()[-300:])
PY
done 2>&1 | tee $OUT/summary.txt
