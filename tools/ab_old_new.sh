#!/bin/bash
# A/B on ONE box: the C2 line of the previous commit's library (tools/_ab/old: `git archive <commit> compress_amd bench.py include |
# tar -x -C tools/_ab/old`, built there) against the working tree's, alternating.  Usage (GPU box, repo root): bash tools/ab_old_new.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab
mkdir -p $OUT
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 6 --warmup 2"
cd $R
timeout 300 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "probe_rounds or fastest_epoch or corpus_units or stress_mixes_bit or raw_only or ragged or edge" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -2 $OUT/pytest_subset.log
for k in 1 2; do
  for w in old new newh; do
    if [ $w = old ]; then cd $R/tools/_ab/old; else cd $R; fi
    E="KC_X=0"
    BB="$B"
    if [ $w = newh ]; then BB="--config C2H --no-also --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2"; fi
    env $E timeout 300 python bench.py $BB > $OUT/${w}_$k.json 2> $OUT/${w}_$k.err
    python - <<PY
import json
j = json.loads(open("$OUT/${w}_$k.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("$w $k", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "prep", r.get("table_prep_ms"), "entropy", r.get("entropy_kernel_ms"), "pipeline", r.get("pipeline_kernel_ms"))
PY
  done
done
