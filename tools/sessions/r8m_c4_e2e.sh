#!/bin/bash
# Session r8m: C4 host-buffer path on the tree with the halves rule built in: default against the explicit sizes, same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8m}
mkdir -p $OUT
cd $R
ulimit -c 0
for mib in 0 512 1024 0 512 1024; do
  E="KC_X=0"; [ $mib != 0 ] && E="KC_HOST_ROLL_MIB=$mib"
  env $E timeout 400 python bench.py --config C4 --no-also --no-cpu-baseline --no-floor --no-device-verify --steps 5 --warmup 2 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); e = j.get("end_to_end") or {}
    print("C4 sub-batch $mib MiB |", j["value"], "MB/s device-resident | e2e", e.get("value"), e.get("ms_per_batch"), "single", (e.get("single_call") or {}).get("value"), e.get("error"), "pcie", (e.get("pcie_ceiling") or {}).get("h2d_GBps"))
except Exception as ex:
    print("C4 $mib FAILED", ex, open("$OUT/run.err").read()[-300:])
PY
done
