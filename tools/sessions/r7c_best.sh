#!/bin/bash
# Session r7c: s2.EncodeBest: 6 / 7 / 8 waves per SIMD at the batch that is exactly one residency for each (1.5 / 1.75 / 2 GiB).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7c
mkdir -p $OUT
cd $R
ulimit -c 0
for cfg in "sbw6 1.5" "sbw7 1.75" "sbw8 2.0" "sbw6 1.0" "sbw6 3.0" "sbw7 1.75" "sbw8 2.0"; do
  set -- $cfg
  KC_MAX_SCRATCH_MIB=230000 KC_LIB_TAG=$1 timeout 300 python bench.py --config C4 --s2-level 4 --gib $2 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1 $2 GiB', j['value'], 'MB/s', j['ms_per_step'], 'ms/step roundtrip', j['device_roundtrip_all_frames'])" | tee -a $OUT/summary.txt
done
