#!/bin/bash
# Round 3, second session, second probe: the empty-group filter, the deeper checksum-and-copy loop, one clear launch, cached layout
# uploads.  Usage (GPU box, repo root): bash tools/r3c_probe.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "epoch or skip_segments or raw_only or corpus_units or stress or edge or ragged or streams_bit" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
B="--no-also --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2"
run() { # tag, config, env...
  local tag=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg $B > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print("$tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "prep", r.get("table_prep_ms"), "entropy", r.get("entropy_kernel_ms"),
          "pipeline", r.get("pipeline_kernel_ms"), "frac", r.get("frac"), "ratio", j.get("ratio"), "verify", j.get("device_roundtrip_all_frames"))
except Exception as e:
    print("$tag failed", e)
PY
}
run C2_default C2 KC_X=0
run C2_nofilter C2 KC_ZFAST_FILTER=0
run C2H_default C2H KC_X=0
run C2H_nofilter C2H KC_ZFAST_FILTER=0
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_C2H -- python bench.py --config C2H --steps 3 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify > $OUT/kt_C2H.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
f = glob.glob(os.path.join(out, "kt_C2H", "**", "*.db"), recursive=True)
if f:
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_C2H.csv"), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
    print(open(os.path.join(out, "kernel_stats_C2H.csv")).read())
PY
rm -rf $OUT/kt_C2H
tail -3 $OUT/pytest_gpu.log
