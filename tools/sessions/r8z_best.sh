#!/bin/bash
# Session r8z: s2.EncodeBest (24 576 blocks = one residency at 4 blocks per wave, 6 waves per SIMD) as two launches on three contexts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r8z
mkdir -p $OUT
cd $R
ulimit -c 0
one() {
  lab=$1; shift
  timeout 600 python bench.py --config C4 --s2-level 4 --gib 1.5 --no-also --no-end-to-end --no-floor "$@" 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "parity", j.get("bit_exact_vs_oracle_on_sample"), j.get("device_roundtrip_all_frames"), "ctx", j.get("contexts"), "split", j.get("split"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-500:])
PY
}
one "s2.EncodeBest one launch per step" --no-cpu-baseline --no-device-verify --steps 4 --warmup 1
one "s2.EncodeBest two launches, 3 contexts" --pipeline --contexts 3 --split 2 --steps 6 --warmup 2
one "s2.EncodeBest one launch per step" --no-cpu-baseline --no-device-verify --steps 4 --warmup 1
one "s2.EncodeBest two launches, 3 contexts" --pipeline --contexts 3 --split 2 --no-cpu-baseline --no-device-verify --steps 6 --warmup 2
