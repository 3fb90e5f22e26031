#!/bin/bash
# Session r7g: entropy stage, the gather's long runs loaded beside the step's short runs (base) vs one round trip each (lpre0); C2 one context;
# wave 0's clocks per gather sub-step in both forms (fine / fine0); parity subset first.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7g
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/pytest_subset timeout 900 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "corpus_units or edge or stress or ragged or raw_only or rle_literal or long_units or randomized_options or parse_matches"; echo "pytest rc $? $(tail -1 $OUT/pytest_subset.log)" | tee $OUT/summary.txt
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 5 --warmup 2 --no-pipeline"
for tag in lpre0 base lpre0 base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 300 python bench.py $B 2>$OUT/$tag.err | tail -1 > $OUT/$tag.json
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"), "bit_exact", j.get("bit_exact_vs_oracle_on_sample"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
done
B1="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify --steps 1 --warmup 1 --no-pipeline"
for tag in fine0 fine; do
    env KC_LIB_TAG=$tag KC_K2_PROF=1 timeout 150 python bench.py $B1 > $OUT/$tag.json 2> $OUT/$tag.err
    echo "== $tag" | tee -a $OUT/summary.txt
    grep "K2 " $OUT/$tag.err | tail -3 | cut -c1-700 | tee -a $OUT/summary.txt
done
# the two-context default line
for tag in lpre0 base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 300 python bench.py --config C2 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$tag two contexts', j['value'], 'MB/s', j['ms_per_step'], 'ms/step')" | tee -a $OUT/summary.txt
done
