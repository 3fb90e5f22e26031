#!/bin/bash
# Session r5k: speculation policy of the SpeedDefault match finder with the LDS source ring in place (C3, 4 GiB): width after a match /
# growth on a miss (KC_SPEC_W0 / KC_SPEC_GROW: 0 keep, 1 +1, 2 double).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5k
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C3 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 3 --warmup 1 --no-pipeline"
for pol in "2 2" "1 1" "2 1" "1 2" "3 1" "2 0" "3 0" "4 0"; do
  set -- $pol
  KC_SPEC_W0=$1 KC_SPEC_GROW=$2 timeout 200 python bench.py $B > $OUT/p_$1_$2.json 2> $OUT/p_$1_$2.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/p_$1_$2.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("w0=$1 grow=$2:", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"))
except Exception as e:
    print("w0=$1 grow=$2 FAILED", e)
PY
done
