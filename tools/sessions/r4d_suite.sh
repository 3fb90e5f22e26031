#!/bin/bash
# Round 4: the whole GPU suite (incl. the device against the translated Go reference and its committed hashes), the default bench line,
# kernel-trace stats of C2 and C2H.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -8 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-400
for cfg in C2 C2H; do
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kt_$cfg -- python bench.py --config $cfg --steps 4 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_$cfg.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
f = glob.glob(os.path.join(out, "kt_$cfg", "**", "*.db"), recursive=True)
if f:
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_$cfg.csv"), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
    print(open(os.path.join(out, "kernel_stats_$cfg.csv")).read()[:700])
PY
rm -rf $OUT/kt_$cfg
done
