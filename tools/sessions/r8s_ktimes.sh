#!/bin/bash
# Session r8s: per-launch durations of the dominant kernel under the closing defaults (two launches per step for C2 / C3 / C4, unchained match finders for C5): HIP events inside bench.py beside rocprofv3's kernel trace of the same run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8s}
mkdir -p $OUT
cd $R
ulimit -c 0
B="--no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify"
for c in C2 C3 C4 C5; do
  timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_$c -- python bench.py --config $c $B --steps 8 --warmup 3 > $OUT/kt_$c.log 2>&1
  python - <<PY | tee -a $OUT/summary.txt
import sqlite3, glob, os, json
f = glob.glob(os.path.join("$OUT", "kt_$c", "**", "*.db"), recursive=True)
k = sqlite3.connect(f[0])
rows = list(k.execute("select name, start, end from kernels order by start"))
mf = [r for r in rows if ("match_grp" in r[0] or "kc_s2_encode_kernel" in r[0])]
t0 = mf[0][1]
print("$c rocprof match finder launches (start, end, ms):", [(round((r[1] - t0) / 1e6, 1), round((r[2] - t0) / 1e6, 1), round((r[2] - r[1]) / 1e6, 1)) for r in mf])
j = json.loads([l for l in open("$OUT/kt_$c.log") if l.startswith("{")][-1]); r = j["roofline"]
print("$c bench under rocprof:", j["value"], j["ms_per_step"], "kernel_ms", r["kernel_ms"], "steps", r.get("kernel_ms_steps"))
PY
  rm -rf $OUT/kt_$c
  timeout 300 python bench.py --config $c $B --steps 8 --warmup 3 2>$OUT/run.err | tail -1 > $OUT/run_$c.json
  python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/run_$c.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("$c bench alone:", j["value"], j["ms_per_step"], "kernel_ms", r["kernel_ms"], "steps", r.get("kernel_ms_steps"), "achieved", r["achieved"], r.get("achieved_launches_in_flight"))
PY
done
