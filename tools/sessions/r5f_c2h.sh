#!/bin/bash
# Session r5f (GPU box, repo root): C2H (4 GiB high-entropy input) with the checksum-and-copy kernel at eight loads per lane in flight
# (KC_XXH_FIN_MODE=3) against four (1, round 4's default) and the LDS-ring aligned stores (2); kernel stats of mode 3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5f
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "raw_only or prescan" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_subset.log)"
B="--config C2H --no-also --no-cpu-baseline --no-end-to-end --steps 8 --warmup 3 --no-pipeline"
for m in 1 3 2 1 3; do
    KC_XXH_FIN_MODE=$m timeout 300 python bench.py $B > $OUT/mode$m.json 2> $OUT/mode$m.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/mode$m.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("mode $m", j["value"], "MB/s", j["ms_per_step"], "ms/step; pipeline kernels", r.get("pipeline_kernel_ms"), "frac", r.get("frac"), "read-only", r.get("read_only_frac"), "roundtrip", j.get("device_roundtrip_all_frames"))
except Exception as e:
    print("mode $m FAILED", e, open("$OUT/mode$m.err").read()[-300:])
PY
done 2>&1 | tee $OUT/summary.txt
KC_XXH_FIN_MODE=3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o c2h --output-format csv -- python bench.py $B --no-device-verify --steps 4 > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_C2H_mode3.csv && head -12 $f | cut -c1-160
find $OUT/prof -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
