#!/bin/bash
# Session r8h: C3's host-buffer path (8 lanes, 1 GiB sub-batches, match finders unchained) is FASTER than its device-resident arrangement
# (4 GiB per launch, two contexts, one match finder at a time): 17.1 vs 14.6 GB/s on the r8f box.  Is a 4 GiB batch better served as
# four 1 GiB launches side by side?  bench.py at 1 GiB per step with 4 / 5 contexts and 4 match finders in flight, against the default.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8h}
mkdir -p $OUT
cd $R
ulimit -c 0
one() {  # label, flags
  lab=$1; shift
  timeout 500 python bench.py --no-also --no-cpu-baseline --no-floor --no-device-verify --no-end-to-end "$@" 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "ctx", j.get("contexts"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-300:])
PY
}
for rep in 1 2; do
  one "C3 4 GiB default" --config C3 --steps 6 --warmup 3
  one "C3 2 GiB x 3 ctx / 2 mf" --config C3 --gib 2 --contexts 3 --mf-in-flight 2 --steps 12 --warmup 6
  one "C3 2 GiB x 4 ctx / 3 mf" --config C3 --gib 2 --contexts 4 --mf-in-flight 3 --steps 12 --warmup 8
  one "C3 2 GiB x 4 ctx / 2 mf" --config C3 --gib 2 --contexts 4 --mf-in-flight 2 --steps 12 --warmup 8
  one "C2 4 GiB default" --config C2 --steps 8 --warmup 3
  one "C2 2 GiB x 3 ctx / 2 mf" --config C2 --gib 2 --contexts 3 --mf-in-flight 2 --steps 16 --warmup 6
  one "C2 2 GiB x 4 ctx / 3 mf" --config C2 --gib 2 --contexts 4 --mf-in-flight 3 --steps 16 --warmup 8
  one "C5 1 GiB default (3 ctx / 2 mf)" --config C5 --steps 12 --warmup 6
  one "C5 0.5 GiB x 5 ctx / 4 mf" --config C5 --gib 0.5 --contexts 5 --mf-in-flight 4 --steps 24 --warmup 10
done
