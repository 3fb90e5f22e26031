#!/bin/bash
# Session r7f: s2.EncodeBest with the buckets of s+1 looked up a pass ahead (base) vs not (sbnopre), 1.5 GiB; parity first.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7f
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/pytest_best timeout 900 python -m pytest tests/test_gpu_s2.py -m gpu -q -x -k "best or writer or stream"; echo "pytest rc $? $(tail -1 $OUT/pytest_best.log)" | tee $OUT/summary.txt
for tag in sbnopre base sbnopre base; do
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  for lvl in 4 5; do
  env $E timeout 300 python bench.py --config C4 --s2-level $lvl --gib 1.5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag level $lvl', j['value'], 'MB/s', j['ms_per_step'], 'ms/step roundtrip', j['device_roundtrip_all_frames'], 'ratio', j['ratio'])" | tee -a $OUT/summary.txt
  done
done
