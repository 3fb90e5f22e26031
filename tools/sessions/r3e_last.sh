#!/bin/bash
# Round 3, second session, last call: the default bench line and the kernel-trace stats of C2 on the final code, then the GPU suite of
# everything that touches the zstd path (the S2 files are unchanged since the full run of tools/r3d_final.sh).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/last_r03b
mkdir -p $OUT
cd $R
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kt_C2 -- python bench.py --steps 4 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_C2.log 2>&1
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
    print(open(os.path.join(out, "kernel_stats_C2.csv")).read()[:900])
PY
rm -rf $OUT/kt_C2
timeout 260 python -m pytest tests -x -q -m gpu --ignore tests/test_gpu_s2.py --ignore tests/test_ref_s2asm.py --ignore tests/test_reference_amd64_golden.py > $OUT/pytest_gpu_zstd.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_zstd.log
tail -3 $OUT/pytest_gpu_zstd.log
