#!/bin/bash
# Session r5m: C2 with one, two and three contexts in flight (bench.py --contexts): does the entropy stage of step i drain under the
# match finders of steps i+1 and i+2?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5m
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 12 --warmup 3"
for n in 1 2 3 4 2 3; do
    if [ $n = 1 ]; then A="--no-pipeline"; else A="--pipeline --contexts $n"; fi
    timeout 300 python bench.py $B $A > $OUT/ctx$n.json 2> $OUT/ctx$n.err
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/ctx$n.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("contexts $n:", j["value"], "MB/s", j["ms_per_step"], "ms/step (median", j.get("ms_per_step_median"), "); kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"))
except Exception as e:
    print("contexts $n FAILED", e, open("$OUT/ctx$n.err").read()[-300:])
PY
done
