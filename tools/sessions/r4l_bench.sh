#!/bin/bash
# Round 4: the default bench line on the final code (two contexts), its one-context twin on the same box, kernel-trace stats of both for C2.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4l
mkdir -p $OUT
cd $R
timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-260
timeout 200 python bench.py --no-pipeline --no-also --no-cpu-baseline --no-end-to-end > $OUT/bench_one_context.json 2> $OUT/bench_one_context.err
tail -1 $OUT/bench_one_context.json | cut -c1-260
for mode in pipe nopipe; do
F=""; [ $mode = nopipe ] && F="--no-pipeline"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_$mode -- python bench.py $F --steps 4 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_$mode.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
f = glob.glob(os.path.join(out, "kt_$mode", "**", "*.db"), recursive=True)
if f:
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_C2_$mode.csv"), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
    print(open(os.path.join(out, "kernel_stats_C2_$mode.csv")).read()[:500])
PY
rm -rf $OUT/kt_$mode
done
timeout 300 python -m pytest tests/test_gpu_s2.py tests/test_gpu_zstd.py -x -q -m gpu -k "uncompressed or writer or above_one_mib or stream_framing" 2>&1 | tail -3
