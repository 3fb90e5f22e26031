#!/bin/bash
# Round 4, first GPU call on product code: the no-match pre-scan (kc_zstd_prescan.hip) — its parity tests, C2H with it, C2 beside it.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4a
mkdir -p $OUT
cd $R
timeout 280 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "prescan or raw_only_frames or probe_rounds or fastest_epoch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 200 python bench.py --config C2H --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/bench_C2H.json 2> $OUT/bench_C2H.err
tail -1 $OUT/bench_C2H.json | cut -c1-400
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_C2H -- python bench.py --config C2H --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_C2H.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
for tag in ("C2H",):
    f = glob.glob(os.path.join(out, "kt_" + tag, "**", "*.db"), recursive=True)
    if f:
        k = sqlite3.connect(f[0])
        rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        with open(os.path.join(out, "kernel_stats_%s.csv" % tag), "w") as fo:
            w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
            for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
        print(open(os.path.join(out, "kernel_stats_%s.csv" % tag)).read()[:1500])
PY
rm -rf $OUT/kt_C2H
timeout 240 python bench.py --steps 6 --warmup 2 --no-also --no-cpu-baseline --no-end-to-end > $OUT/bench_C2.json 2> $OUT/bench_C2.err
tail -1 $OUT/bench_C2.json | cut -c1-400
