#!/bin/bash
# Session r7b: s2.EncodeBest at 1.5 GiB (24 576 blocks): waves per SIMD 4 (base) / 5 / 6 — more blocks resident against spills.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7b
mkdir -p $OUT
cd $R
ulimit -c 0
for tag in base sbw5 sbw6 base sbw5 sbw6; do
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  env $E timeout 300 python bench.py --config C4 --s2-level 4 --gib 1.5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag 1.5GiB', j['value'], 'MB/s', j['ms_per_step'], 'ms/step roundtrip', j['device_roundtrip_all_frames'])" | tee -a $OUT/summary.txt
done
