#!/bin/bash
# Session r7t: entropy kernel compiled for size (eos: -Os 80 KB, eoz: -Oz 67 KB) vs the product (-O3, 112 KB of code against a 64 KB instruction cache per two CUs); C2, one and two contexts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7t
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config ${CONFIG:-C2} --no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify"
for rep in 1 2; do
for tag in ${TAGS:-base eos eoz}; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    for mode in "--no-pipeline --steps 5 --warmup 2" "--steps 8 --warmup 2"; do
    env $E timeout 300 python bench.py $B $mode 2>$OUT/$tag.err | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$tag $mode |', j['value'], 'MB/s', j['ms_per_step'], 'ms/step; kernel', r.get('kernel_ms'), 'entropy', r.get('entropy_kernel_ms'))" | tee -a $OUT/summary.txt
    done
done
done
