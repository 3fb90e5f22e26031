#!/bin/bash
# Round 4, closing campaign on the MI355X: the whole GPU suite, smoke, the default bench line, kernel-trace stats of the configurations
# not profiled yet this round, the counter traffic of the C2H pipeline, the crossover / latency tables and the hook under native callers.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4k
mkdir -p $OUT
cd $R
timeout 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; grep -c "smoke ok" $OUT/smoke.log
timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
for cfg in C3 C4 C5; do
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt_$cfg -- python bench.py --config $cfg --steps 3 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_$cfg.log 2>&1
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
    print(open(os.path.join(out, "kernel_stats_$cfg.csv")).read()[:400])
PY
rm -rf $OUT/kt_$cfg
done
# C2H: HBM bytes of the WHOLE pipeline of one step (FETCH_SIZE and WRITE_SIZE in separate passes; the last dispatch of every kernel)
for ctr in FETCH_SIZE WRITE_SIZE; do
timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$ctr -- python bench.py --config C2H --steps 1 --warmup 2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/pmc_$ctr.log 2>&1
done
python - <<PY
import sqlite3, glob, json, os, collections
out = "$OUT"
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*.db"), recursive=True)
    if not f: continue
    c = sqlite3.connect(f[0])
    cols = [x[1] for x in c.execute("pragma table_info(counters_collection)")]
    ik, ic, iv, idp = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in c.execute("select * from counters_collection"):
        if r[ic] == ctr: per[r[ik][:60]][r[idp]] += float(r[iv])
    res[ctr] = {k: v[max(v)] * 1024.0 for k, v in per.items() if k.startswith(("kc_", "void kc_"))}  # the last dispatch of each kernel = the timed step
json.dump(res, open(os.path.join(out, "pmc_C2H_pipeline.json"), "w"), indent=1)
tot = sum(sum(v.values()) for v in res.values())
print("C2H pipeline HBM bytes (last dispatch of every kernel):", tot, {k: round(sum(v.values()) / 1e9, 3) for k, v in res.items()})
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
timeout 500 python tools/crossover.py --out $OUT --max-units 8192 > $OUT/crossover.log 2>&1; tail -5 $OUT/crossover.log
g++ -O2 -std=c++17 -I include tools/hook_bench.cpp -o /tmp/hook_bench -L compress_amd -lkcgpu -Wl,-rpath,$PWD/compress_amd -lpthread
timeout 120 /tmp/hook_bench 4 | tee $OUT/hook_bench.json
timeout 200 python tools/lds_lat.py --out $OUT/lds_lat.json --modes 0,1 > $OUT/lds_lat.log 2>&1; grep -c mode $OUT/lds_lat.log
