#!/bin/bash
# Session r7q: dictionary-table broadcast of C5: 16 KiB per workgroup in registers + non-temporal stores (base) vs one word per thread (bc0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7q
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "dictionary" > $OUT/pytest.log 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify"
for rep in 1 2 3; do
for tag in bc0 base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    for mode in "--no-pipeline --steps 10 --warmup 3" "--steps 12 --warmup 6"; do
    env $E timeout 300 python bench.py $B $mode 2>$OUT/$tag.err | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$tag $mode |', j['value'], 'MB/s', j['ms_per_step'], 'ms/step; kernel', r.get('kernel_ms'), 'prep', r.get('table_prep_ms'), 'entropy', r.get('entropy_kernel_ms'))" | tee -a $OUT/summary.txt
    done
done
done
