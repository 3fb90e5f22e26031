#!/bin/bash
# Session r8q: the entropy stage on a high-priority stream of its own (what the rolling pipeline's lanes do) in the device-resident arrangement
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8q}
mkdir -p $OUT
cd $R
ulimit -c 0
one() {  # label, flags
  lab=$1; shift
  timeout 500 python bench.py --no-also --no-cpu-baseline --no-device-verify --no-end-to-end --no-floor "$@" 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "in flight avg", r.get("launches_in_flight_avg"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-400:])
PY
}
for rep in 1 2 3; do
  one "C2 default" --config C2 --steps 14 --warmup 3
  one "C2 stage 2 high priority" --config C2 --stage2-priority 1 --steps 14 --warmup 3
  one "C3 default" --config C3 --steps 8 --warmup 2
  one "C3 stage 2 high priority" --config C3 --stage2-priority 1 --steps 8 --warmup 2
  one "C5 default" --config C5 --steps 12 --warmup 3
  one "C5 stage 2 high priority" --config C5 --stage2-priority 1 --steps 12 --warmup 3
done
