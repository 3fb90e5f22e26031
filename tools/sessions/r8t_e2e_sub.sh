#!/bin/bash
# Session r8t: host-buffer path of C2 / C3 by sub-batch size at four calls in flight (default: a quarter of the call = 1 GiB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8t}
mkdir -p $OUT
cd $R
ulimit -c 0
for c in C2 C3; do
for mib in 0 2048 0 2048; do
  E="KC_X=0"; [ $mib != 0 ] && E="KC_HOST_ROLL_MIB=$mib"
  env $E timeout 500 python bench.py --config $c --no-also --no-cpu-baseline --no-floor --no-device-verify --steps 5 --warmup 2 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); e = j.get("end_to_end") or {}
    print("$c sub-batch $mib MiB |", j["value"], "MB/s device-resident | e2e", e.get("value"), e.get("ms_per_batch"), "single", (e.get("single_call") or {}).get("value"), e.get("error"))
except Exception as ex:
    print("$c $mib FAILED", ex, open("$OUT/run.err").read()[-300:])
PY
done
done
