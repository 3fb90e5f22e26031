#!/bin/bash
# Session r5l: the DPP dependency check in kc_s2_encode_kernel (C4) and kc_zbetter_match_grp_kernel (C5) against round 4's kernels
# (KC_LIB_TAG=olddf), after a GPU parity subset of the S2 and SpeedBetter paths.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5l
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_s2.py tests/test_gpu_zstd.py -x -q -m gpu -k "64k_blocks or edge or large_blocks or better or snappy or corpus_units or framed" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_subset.log)"
for cfg in C4 C5; do
for tag in olddf base olddf base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 300 python bench.py --config $cfg --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 4 --warmup 1 > $OUT/${cfg}_${tag}.json 2> $OUT/${cfg}_${tag}.err
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/${cfg}_${tag}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$cfg $tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"))
except Exception as e:
    print("$cfg $tag FAILED", e, open("$OUT/${cfg}_${tag}.err").read()[-300:])
PY
done
done
