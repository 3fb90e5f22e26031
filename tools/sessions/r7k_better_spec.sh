#!/bin/bash
# Session r7k: SpeedBetter — early offset-2 request (base) vs not (zbp0); speculation width after a match x growth on the fused kernel (C5, one context)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7k
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify --no-pipeline --steps 8 --warmup 3"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$1 |', j['value'], 'MB/s', j['ms_per_step'], 'ms/step; kernel', r.get('kernel_ms'))"; }
for rep in 1 2; do
  for tag in zbp0 base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 300 python bench.py $B 2>/dev/null | line "$tag" | tee -a $OUT/summary.txt
  done
done
for w0 in 2 4 8; do for gr in 1 2; do
  KC_SPEC_W0=$w0 KC_SPEC_GROW=$gr timeout 300 python bench.py $B 2>/dev/null | line "w0=$w0 grow=$gr" | tee -a $OUT/summary.txt
done; done
KC_SPEC_W0=16 KC_SPEC_GROW=0 timeout 300 python bench.py $B 2>/dev/null | line "w0=16 grow=0 (default)" | tee -a $OUT/summary.txt
