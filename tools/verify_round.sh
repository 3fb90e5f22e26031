#!/bin/bash
# Round-end verification on the GPU box (repo root): the GPU suite, smoke(), the default bench line, and the kernel-trace stats of
# the default bench command.  Usage: bash tools/verify_round.sh <tag>
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/verify_$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_C2 -- python bench.py --no-also --no-cpu-baseline --no-end-to-end > $OUT/kt_C2.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
f = glob.glob(os.path.join(out, "kt_C2", "**", "*.db"), recursive=True)
if f:
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_C2.csv"), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
PY
rm -rf $OUT/kt_C2
tail -3 $OUT/pytest_gpu.log; tail -4 $OUT/smoke.log; tail -1 $OUT/bench_default.json | cut -c1-600; head -8 $OUT/kernel_stats_C2.csv
