#!/bin/bash
# Session r4s (GPU box, repo root): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the dominant kernels whose
# source changed since profiles/pmc_traffic.json was stamped; entries of the other configurations are kept.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4s
mkdir -p $OUT
cd $R
for c in ${PMC_CONFIGS:-C2 C3 C4 C5 C4A}; do
  timeout 170 python bench.py --config $c --steps 2 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --no-pipeline --pmc > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  echo "$c rc=$? $(date +%T)"
done
python - <<PY
import json, os
out = "$OUT"
pj = json.load(open(os.path.join("$R", "profiles", "pmc_traffic.json")))
ents = pj["entries"]
for t in ("C2", "C3", "C4", "C4A", "C5"):
    f = os.path.join(out, "pmc_%s.json" % t)
    if not os.path.exists(f):
        continue
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        if r["traffic"] and r["traffic_source"].startswith("measured"):
            algo = int(j["config"]["units_per_gpu"] * j["config"]["unit_bytes"] * (1 + j["ratio"]))
            e = {"config": t, "units": j["config"]["units_per_gpu"], "corpus": j["config"]["corpus"], "unit_bytes": j["config"]["unit_bytes"],
                 "kernel": r["kernel"], "kernel_source_sha16": r["kernel_source_sha16"], "kernel_hbm_bytes": r["traffic"],
                 "algorithmic_bytes": algo, "ratio_to_algorithmic": round(r["traffic"] / algo, 2), "kernel_ms": r["kernel_ms"],
                 "collected": "round 4, session r4s"}
            ents = [x for x in ents if x.get("config") != t] + [e]
            print("pmc", t, e["kernel_hbm_bytes"], e["ratio_to_algorithmic"], e["kernel_source_sha16"])
        else:
            print("pmc", t, "no measurement:", r.get("traffic_source"))
    except Exception as ex:
        print("pmc", t, "failed:", ex)
pj["entries"] = ents
json.dump(pj, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
PY
