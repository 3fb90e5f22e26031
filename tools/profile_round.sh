#!/bin/bash
# Round profile on the GPU box (repo root): one bench line per BASELINE configuration, rocprofv3 kernel-trace stats of the default
# bench command and of the other configurations, and HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of every
# configuration's dominant kernel, stamped with the kernel source hash.  Usage: bash tools/profile_round.sh <tag>
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
# 1. the default command, as the driver runs it (C2, 10 steps)
python bench.py > $OUT/bench_C2.json 2> $OUT/bench_C2.err
# 2. kernel-trace stats of the same command
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_C2 -- python bench.py --no-also --no-cpu-baseline --no-end-to-end > $OUT/kt_C2.log 2>&1
# 3. the other configurations: bench line + kernel-trace stats
for c in C2H C3 C4 C4A C5 B4; do
  python bench.py --config $c --steps 5 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$c -- python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_$c.log 2>&1
done
# 3b. the LDS-table kernels at small batches (one wave per unit): kernel-trace stats of the crossover sweep up to 256 units
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_LDS -- python tools/crossover.py --max-units 256 --reps 1 --out $OUT/lds_sweep > $OUT/kt_LDS.log 2>&1
# 4. HBM traffic of the dominant kernels (bench.py --pmc: one rocprofv3 pass per counter, one step each)
# PMC_CONFIGS: the configurations whose dominant kernel changed since profiles/pmc_traffic.json was stamped (default: all)
for c in ${PMC_CONFIGS:-C2 C3 C4 C4A C5}; do
  python bench.py --config $c --steps 2 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --pmc > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
python - <<PY
import sqlite3, glob, json, csv, os
out = "$OUT"
def stats(tag):
    f = glob.glob(os.path.join(out, "kt_" + tag, "**", "*.db"), recursive=True)
    if not f: return
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_%s.csv" % tag), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
for t in ("C2", "C2H", "C3", "C4", "C4A", "C5", "B4", "LDS"): stats(t)
entries = []
try:  # entries of configurations not measured in this run are kept (bench.py checks the kernel source hash of each)
    old = {e["config"]: e for e in json.load(open(os.path.join("$R", "profiles", "pmc_traffic.json")))["entries"]}
except Exception:
    old = {}
for t in ("C2", "C3", "C4", "C4A", "C5"):
    if not os.path.exists(os.path.join(out, "pmc_%s.json" % t)):
        if t in old: entries.append(old[t])
        continue
    try:
        j = json.loads(open(os.path.join(out, "pmc_%s.json" % t)).read().strip().splitlines()[-1])
        r = j["roofline"]
        if r["traffic"]:
            entries.append({"config": t, "units": j["config"]["units_per_gpu"], "corpus": j["config"]["corpus"], "unit_bytes": j["config"]["unit_bytes"],
                            "kernel": r["kernel"], "kernel_source_sha16": r["kernel_source_sha16"], "kernel_hbm_bytes": r["traffic"],
                            "algorithmic_bytes": int(j["config"]["units_per_gpu"] * j["config"]["unit_bytes"] * (1 + j["ratio"])),
                            "ratio_to_algorithmic": round(r["traffic"] / (j["config"]["units_per_gpu"] * j["config"]["unit_bytes"] * (1 + j["ratio"])), 2),
                            "kernel_ms": r["kernel_ms"]})
    except Exception as e:
        print("pmc", t, "failed:", e)
json.dump({"command": "python bench.py --config <C> --pmc  (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, one step each; raw (FETCH_SIZE+WRITE_SIZE)*1024)",
           "entries": entries}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(entries, indent=1))
PY
find $OUT -name "*.db" -delete
find $OUT -type d -name "kt_*" | xargs rm -rf
for c in C2 C2H C3 C4 C4A C5 B4; do tail -c 3000 $OUT/bench_$c.json | tail -1 | cut -c1-400; done
head -8 $OUT/kernel_stats_C2.csv
