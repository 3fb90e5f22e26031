#!/bin/bash
# Session r8b: C5 with two match finders on the chip together (r8a: 18.5 -> 21.4 GB/s with the product kernel, 22.9 with the kernel held
# to 128 VGPRs).  Here: 96 VGPRs (zbw5), the host-buffer path (8 lanes whose match finders always share the chip) with each build, and
# one batch that is more than a residency by itself (2 / 4 GiB: 16 384 / 32 768 units, one context).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8b}
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C5 --no-also --no-cpu-baseline --no-floor --no-device-verify"
one() {  # tag, label, flags
  tag=$1; shift; lab=$1; shift
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  env $E timeout 400 python bench.py $B "$@" 2>$OUT/$tag.err | tail -1 > $OUT/$tag.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]; e = j.get("end_to_end") or {}
    print("$tag $lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "ctx", j.get("contexts"),
          "| e2e", e.get("value"), e.get("ms_per_batch"), "single", (e.get("single_call") or {}).get("value"), e.get("error"))
except Exception as ex:
    print("$tag $lab FAILED", ex, open("$OUT/$tag.err").read()[-300:])
PY
}
for rep in 1 2; do
  for tag in base zbw4 zbw5; do
    one $tag "3ctx/2mf" --no-end-to-end --contexts 3 --mf-in-flight 2 --steps 12 --warmup 6
  done
done
for tag in base zbw4 zbw5; do
  one $tag "4ctx/3mf" --no-end-to-end --contexts 4 --mf-in-flight 3 --steps 12 --warmup 8
done
for tag in base zbw4 base zbw4; do
  one $tag "e2e" --contexts 3 --mf-in-flight 2 --steps 6 --warmup 4
done
for g in 2 4; do
  for tag in base zbw4 zbw5; do
    one $tag "one context, $g GiB" --no-end-to-end --no-pipeline --gib $g --steps 4 --warmup 2
  done
done
