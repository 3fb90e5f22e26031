#!/bin/bash
# Session r6p: sub-batch size of the rolling pipeline (KC_HOST_ROLL_MIB) on C4 and C2: one call and three calls in flight.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6p
mkdir -p $OUT
cd $R
ulimit -c 0
for mib in 0 256 128; do
  KC_HOST_ROLL_MIB=$mib timeout 300 python tools/e2e_probe.py C4 --ctx 3 --steps 9 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C4 roll_mib=$mib one_call', d['one_call_ms'], 'steady', d['two_contexts_ms_per_batch'], 'dev', d['device_resident_ms'])" | tee -a $OUT/summary.txt
done
for mib in 0 512; do
  KC_HOST_ROLL_MIB=$mib timeout 300 python tools/e2e_probe.py C2 --ctx 3 --steps 9 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C2 roll_mib=$mib one_call', d['one_call_ms'], 'steady', d['two_contexts_ms_per_batch'], 'dev', d['device_resident_ms'])" | tee -a $OUT/summary.txt
done
