#!/bin/bash
# Session r7d: s2.EncodeBetter by batch size (blocks in flight).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7d
mkdir -p $OUT
cd $R
ulimit -c 0
for g in 1.0 2.0 4.0; do
  timeout 300 python bench.py --config C4 --s2-level 1 --gib $g --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('better $g GiB', j['value'], 'MB/s', j['ms_per_step'], 'ms/step roundtrip', j['device_roundtrip_all_frames'])" | tee -a $OUT/summary.txt
done
