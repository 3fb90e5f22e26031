#!/bin/bash
# Session r5n: s2.EncodeBest (one wave per block, 4.5 MiB of tables per block) by blocks in flight: 0.25 / 0.5 / 1 GiB of 64 KiB blocks;
# zstd SpeedBestCompression (B4) at 0.25 / 0.5 GiB.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5n
mkdir -p $OUT
cd $R
ulimit -c 0
for g in 0.25 0.5 1.0; do
  timeout 300 python bench.py --config C4 --s2-level 4 --gib $g --no-also --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 > $OUT/best_$g.json 2> $OUT/best_$g.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/best_$g.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("s2.EncodeBest $g GiB:", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"))
except Exception as e:
    print("best $g FAILED", e, open("$OUT/best_$g.err").read()[-300:])
PY
done
for g in 0.25 0.5; do
  timeout 300 python bench.py --config B4 --gib $g --no-also --no-cpu-baseline --no-end-to-end --steps 2 --warmup 1 > $OUT/b4_$g.json 2> $OUT/b4_$g.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/b4_$g.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("B4 $g GiB:", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"))
except Exception as e:
    print("B4 $g FAILED", e, open("$OUT/b4_$g.err").read()[-300:])
PY
done
