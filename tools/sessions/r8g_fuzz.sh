#!/bin/bash
# Session r8g: device against the translated reference on the closing tree (after the dictionary-prefix kernel's 16-byte stores and the
# host path's sub-batch rule): tools/fuzz_zstd_goref.py, new seeds.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r8g
mkdir -p $OUT
cd $R
ulimit -c 0
for seed in 811 812 813 814 815 816 817 818; do
  timeout 600 python tools/fuzz_zstd_goref.py 400 $seed 2>&1 | tail -12 | tee -a $OUT/fuzz.txt
done
timeout 600 python tools/fuzz_s2_asm.py 2>&1 | tail -8 | tee -a $OUT/fuzz_s2.txt
# C4 host-buffer path by sub-batch size at four calls in flight (default 512 MiB)
for mib in 0 1024 768 0 1024; do
  E="KC_X=0"; [ $mib != 0 ] && E="KC_HOST_ROLL_MIB=$mib"
  env $E timeout 400 python bench.py --config C4 --no-also --no-cpu-baseline --no-floor --no-device-verify --steps 5 --warmup 2 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); e = j.get("end_to_end") or {}
    print("C4 sub-batch $mib MiB |", j["value"], "MB/s device-resident | e2e", e.get("value"), e.get("ms_per_batch"), "single", (e.get("single_call") or {}).get("value"), e.get("error"))
except Exception as ex:
    print("C4 $mib FAILED", ex, open("$OUT/run.err").read()[-300:])
PY
done
