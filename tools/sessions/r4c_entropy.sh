#!/bin/bash
# Round 4: the entropy kernel at 5 workgroups per CU (SEQ_CHUNK 768 brings its LDS under 32 KiB; <= 96 VGPRs) against 4.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4c
mkdir -p $OUT
cd $R
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 5 --warmup 2"
for tag in base w4c768 w5c768 w5c512 base; do
    E=""; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 200 python bench.py $B > $OUT/${tag}.json 2> $OUT/${tag}.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/${tag}.err").read()[-400:])
PY
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
j = json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("default:", j["value"], j["ms_per_step"], j["ms_per_step_spread"], j["roofline"]["frac"])
for k, v in j.get("also", {}).items():
    print(" also", k, v.get("value"), v.get("ms_per_step"), v.get("ms_per_step_spread"), (v.get("roofline") or {}).get("frac"), v.get("error"), v.get("wall_s"))
PY
