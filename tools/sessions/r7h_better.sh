#!/bin/bash
# Session r7h: SpeedBetter match finder with the candidate loads of a probe round / the lazy lookup / the re-search fused into one trip each,
# backward-extension bytes and the offset-2 candidate requested early (base) vs one dependent trip each (zbf0: -DZB_FUSE=0); C5; parity first.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r7h}
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/pytest_subset timeout 1200 python -m pytest tests/test_gpu_zstd.py tests/test_dict_streams.py tests/test_jobs.py -x -q -m gpu -k "${PYTEST_K:-dictionary or corpus_units or edge or stress or ragged or long_units or randomized_options or parse_matches or jobs or streams}"; echo "pytest rc $? $(tail -1 $OUT/pytest_subset.log)" | tee $OUT/summary.txt
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --steps 6 --warmup 2"
for tag in ${TAGS:-zbf0 base zbf0 base}; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    for mode in --no-pipeline ""; do
    env $E timeout 300 python bench.py $B $mode 2>$OUT/$tag.err | tail -1 > $OUT/$tag.json
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag $mode", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"), "ratio", j.get("ratio"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
    done
done
