#!/bin/bash
# Session r5c (GPU box, repo root): (1) wave-role rotation of the entropy kernel (product) against the same kernel without it
# (KC_LIB_TAG=rot0), one context; (2) the gather's sub-steps on wave 0 (KC_LIB_TAG=fine: -DKC_K2_FINE, KC_K2_PROF=1).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c
mkdir -p $OUT
cd $R
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --steps 5 --warmup 2 --no-pipeline"
for tag in rot0 base rot0 base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
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
KC_LIB_TAG=fine KC_K2_PROF=1 timeout 150 python bench.py $B --no-device-verify --steps 1 --warmup 1 > $OUT/fine.json 2> $OUT/fine.err
grep "K2 " $OUT/fine.err | tail -3 | tee -a $OUT/summary.txt
