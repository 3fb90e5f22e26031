#!/bin/bash
# Round 3, second session: round-end verification (GPU suite, smoke, default bench line, kernel-trace stats of C2), the HBM traffic of
# the changed C2 kernel, and C2H with the checksum-and-copy kernel's three store schedules.  Usage (GPU box, repo root): bash tools/r3d_final.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/verify_round.sh r03b
OUT=$R/gpurun_out/verify_r03b
python bench.py --config C2 --steps 2 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --pmc > $OUT/pmc_C2.json 2> $OUT/pmc_C2.err
tail -1 $OUT/pmc_C2.json | cut -c1-200
B="--no-also --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2"
for v in 1 0 2; do
  KC_XXH_FIN_MODE=$v timeout 300 python bench.py --config C2H $B > $OUT/C2H_mode$v.json 2> $OUT/C2H_mode$v.err
  python - <<PY
import json
j = json.loads(open("$OUT/C2H_mode$v.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("C2H mode=$v", j["value"], "MB/s", j["ms_per_step"], "ms/step pipeline", r["pipeline_kernel_ms"], "frac", r["frac"], "verify", j["device_roundtrip_all_frames"])
PY
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_C2H -- python bench.py --config C2H --steps 3 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_C2H.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
f = glob.glob(os.path.join(out, "kt_C2H", "**", "*.db"), recursive=True)
if f:
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_C2H.csv"), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
    print(open(os.path.join(out, "kernel_stats_C2H.csv")).read())
PY
rm -rf $OUT/kt_C2H
