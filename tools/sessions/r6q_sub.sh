#!/bin/bash
# Session r6q: larger sub-batches of the rolling pipeline: C4 at 1 GiB, C2 at 2 GiB; four calls in flight for C4.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6q
mkdir -p $OUT
cd $R
ulimit -c 0
for cfg in "C4 1024 3" "C4 0 4" "C4 1024 4" "C2 2048 3" "C2 0 4"; do
  set -- $cfg
  KC_HOST_ROLL_MIB=$2 timeout 300 python tools/e2e_probe.py $1 --ctx $3 --steps 12 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 roll_mib=$2 calls=$3 one_call', d['one_call_ms'], 'steady', d['two_contexts_ms_per_batch'], 'dev', d['device_resident_ms'])" | tee -a $OUT/summary.txt
done
