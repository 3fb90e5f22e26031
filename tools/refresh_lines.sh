#!/bin/bash
# After tools/profile_round.sh: (re)measure the HBM traffic of the configurations named in PMC_CONFIGS, merge it into
# profiles/pmc_traffic.json, then print one bench line per configuration with the stamped traffic in it.
# Usage (GPU box, repo root): PMC_CONFIGS="C4" bash tools/refresh_lines.sh <tag>
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lines_$TAG
mkdir -p $OUT
cd $R
for c in ${PMC_CONFIGS:-}; do
  python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-device-verify --pmc > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
python - <<PY
import json, os
out, R = "$OUT", "$R"
p = os.path.join(R, "profiles", "pmc_traffic.json")
doc = json.load(open(p))
ent = {e["config"]: e for e in doc["entries"]}
for t in "${PMC_CONFIGS:-}".split():
    try:
        j = json.loads(open(os.path.join(out, "pmc_%s.json" % t)).read().strip().splitlines()[-1])
        r = j["roofline"]
        alg = int(j["config"]["units_per_gpu"] * j["config"]["unit_bytes"] * (1 + j["ratio"]))
        ent[t] = {"config": t, "units": j["config"]["units_per_gpu"], "corpus": j["config"]["corpus"], "unit_bytes": j["config"]["unit_bytes"],
                  "kernel": r["kernel"], "kernel_source_sha16": r["kernel_source_sha16"], "kernel_hbm_bytes": r["traffic"],
                  "algorithmic_bytes": alg, "ratio_to_algorithmic": round(r["traffic"] / alg, 2), "kernel_ms": r["kernel_ms"]}
    except Exception as e:
        print("pmc", t, "failed:", e)
doc["entries"] = [ent[k] for k in ("C2", "C3", "C4", "C5") if k in ent]
json.dump(doc, open(p, "w"), indent=1)
json.dump(doc, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
PY
python bench.py > $OUT/bench_C2.json 2> $OUT/bench_C2.err
for c in C2H C3 C4 C5; do
  python bench.py --config $c --steps 5 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
for c in C2 C2H C3 C4 C5; do tail -1 $OUT/bench_$c.json | cut -c1-300; done
