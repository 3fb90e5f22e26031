#!/bin/bash
# Session r8c: (1) does a second match finder on the chip help C2 / C3 as it helps C5 (r8a / r8b)?  Their kernels are one residency per
# batch (16 workgroups per CU, LDS allows 18), so the next batch's can only start in the tail.  (2) C5 host-buffer path by sub-batch size.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8c}
mkdir -p $OUT
cd $R
ulimit -c 0
one() {  # label, env, flags
  lab=$1; shift; E=$1; shift
  env $E timeout 500 python bench.py --no-also --no-cpu-baseline --no-floor --no-device-verify "$@" 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); r = j["roofline"]; e = j.get("end_to_end") or {}
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "ctx", j.get("contexts"),
          "| e2e", e.get("value"), e.get("ms_per_batch"), "single", (e.get("single_call") or {}).get("value"), e.get("error"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-300:])
PY
}
for rep in 1 2; do
  one "C2 default (2 ctx, 1 mf)" KC_X=0 --config C2 --no-end-to-end --steps 12 --warmup 4
  one "C2 3 ctx, 2 mf" KC_X=0 --config C2 --no-end-to-end --contexts 3 --mf-in-flight 2 --steps 12 --warmup 6
  one "C3 default (2 ctx, 1 mf)" KC_X=0 --config C3 --no-end-to-end --steps 8 --warmup 4
  one "C3 3 ctx, 2 mf" KC_X=0 --config C3 --no-end-to-end --contexts 3 --mf-in-flight 2 --steps 9 --warmup 6
done
for mib in 0 512 128 1024; do
  E="KC_X=0"; [ $mib != 0 ] && E="KC_HOST_ROLL_MIB=$mib"
  one "C5 e2e sub-batch $mib MiB" $E --config C5 --contexts 3 --mf-in-flight 2 --steps 6 --warmup 4
done
