#!/bin/bash
# Session r5e (GPU box, repo root): generic A/B of measurement builds against the product on C2, one context (TAGS="a b ...", default
# the match finder's dependency check over ds_bpermute: depshfl), after a GPU parity subset of the product.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r5e}
mkdir -p $OUT
cd $R
ulimit -c 0
if [ -z "$NO_PYTEST" ]; then
timeout 900 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "${PYTEST_K:-corpus_units or edge or stress or ragged or raw_only or rle_literal or long_units or randomized_options or parse_matches}" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_subset.log)"
fi
B="--config ${CONFIG:-C2} --no-also --no-cpu-baseline --no-end-to-end --steps 5 --warmup 2 --no-pipeline ${BENCH_EXTRA}"
for tag in ${TAGS:-depshfl base depshfl base}; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 300 python bench.py $B > $OUT/${tag}.json 2> $OUT/${tag}.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "pipeline", r.get("pipeline_kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"), "ratio", j.get("ratio"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/${tag}.err").read()[-300:])
PY
done 2>&1 | tee $OUT/summary.txt
