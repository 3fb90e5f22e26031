#!/bin/bash
# Kernel-trace stats of the non-headline BASELINE configs (C3 default, C4 S2, C5 better+dict) on 1 GiB each.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_configs
mkdir -p $OUT
cd $R
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -- python tools/config_bench.py 1 > $OUT/configs.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
f = glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True)
k = sqlite3.connect(f[0])
rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
    w = csv.writer(fh); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
    for r in rows: w.writerow([r[0], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
print(open(os.path.join(out, "kernel_stats.csv")).read())
PY
find $OUT -name "*.db" -delete
grep "^{" $OUT/configs.log
