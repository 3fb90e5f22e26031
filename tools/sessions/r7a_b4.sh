#!/bin/bash
# Session r7a: SpeedBestCompression (B4) by table slots = units in flight (KC_OPT_BEST_SLOTS; 34 MiB each): 2048 (default) / 4096 / 6144.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7a
mkdir -p $OUT
cd $R
ulimit -c 0
for cfg in "2048 0.25" "2048 0.5" "4096 0.5" "6144 0.75" "4096 1.0"; do
  set -- $cfg
  KC_BEST_SLOTS=$1 timeout 400 python bench.py --config B4 --gib $2 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('slots $1 gib $2', j['value'], 'MB/s', j['ms_per_step'], 'ms/step roundtrip', j['device_roundtrip_all_frames'], 'ratio', j['ratio'])" | tee -a $OUT/summary.txt
done
