#!/bin/bash
# Session r5d (GPU box, repo root): where the entropy kernel's time goes, wave 0's clocks per sub-step (measurement builds with a full
# s_waitcnt in front of every mark: KC_LIB_TAG=fine the gather, fine2 the phases behind it), and the phase shares of the product build
# now that the marks are summed in LDS instead of one global atomic each.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5d
mkdir -p $OUT
cd $R
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 1 --warmup 1 --no-pipeline"
for tag in base w3 fine fine2 base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E KC_K2_PROF=1 timeout 150 python bench.py $B > $OUT/$tag.json 2> $OUT/$tag.err
    echo "== $tag: $(python -c "import json;j=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1]);print('entropy', j['roofline']['entropy_kernel_ms'], 'ms')")" | tee -a $OUT/summary.txt
    grep "K2 " $OUT/$tag.err | tail -3 | cut -c1-700 | tee -a $OUT/summary.txt
done
