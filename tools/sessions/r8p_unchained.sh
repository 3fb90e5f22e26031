#!/bin/bash
# Session r8p: two launches per step with the match finders UNCHAINED (a third launch may take wave slots as the oldest one's workgroups leave,
# as the S2 arrangement does) against the chain of lag two (the default); uneven halves.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8p}
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
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "in flight avg", r.get("launches_in_flight_avg"), "ctx", j.get("contexts"), "split", j.get("split"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-400:])
PY
}
for rep in 1 2; do
  one "C2 default (3 ctx, 2 launches, chain 2)" --config C2 --steps 14 --warmup 3
  one "C2 3 ctx, 2 launches, unchained" --config C2 --contexts 3 --split 2 --mf-in-flight 3 --steps 14 --warmup 3
  one "C2 4 ctx, 2 launches, chain 3" --config C2 --contexts 4 --split 2 --mf-in-flight 3 --steps 14 --warmup 3
  one "C3 default" --config C3 --steps 8 --warmup 2
  one "C3 3 ctx, 2 launches, unchained" --config C3 --contexts 3 --split 2 --mf-in-flight 3 --steps 8 --warmup 2
  one "C5 default (3 ctx, chain 2)" --config C5 --steps 12 --warmup 3
  one "C5 3 ctx unchained" --config C5 --contexts 3 --mf-in-flight 3 --steps 12 --warmup 3
done
