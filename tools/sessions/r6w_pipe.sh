#!/bin/bash
# Session r6w: do C3 / C5 / C4-better gain from the two-context pipeline the C2 line uses (match finder of step i+1 under the entropy stage of step i)?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6w
mkdir -p $OUT
cd $R
ulimit -c 0
for c in C3 C5; do
for p in "--no-pipeline" "--pipeline" "--no-pipeline" "--pipeline" "--pipeline --contexts 3"; do
  timeout 300 python bench.py --config $c --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 6 --warmup 2 $p 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$c $p', j['value'], 'MB/s', j['ms_per_step'], 'ms/step; kernel', r['kernel_ms'], 'entropy', r['entropy_kernel_ms'], 'roundtrip', j['device_roundtrip_all_frames'])" | tee -a $OUT/summary.txt
done
done
