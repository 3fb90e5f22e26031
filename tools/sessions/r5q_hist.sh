#!/bin/bash
# Session r5q: the entropy kernel with two literal-histogram copies per wave (by lane parity; KC_LIB_TAG=hist2: -DKC_HCOPIES=2, +4 KiB LDS) against
# one; GPU parity subset on both; LDS conflict counters of both.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5q
mkdir -p $OUT
cd $R
ulimit -c 0
for tag in base hist2; do
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  env $E timeout 600 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "corpus_units or edge or stress or ragged or rle_literal or long_units or randomized_options" > $OUT/pytest_$tag.log 2>&1; echo "$tag pytest rc=$? $(tail -1 $OUT/pytest_$tag.log)"
done
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 5 --warmup 2 --no-pipeline"
for tag in base hist2 base hist2; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 300 python bench.py $B > $OUT/${tag}.json 2> $OUT/${tag}.err
    python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/${tag}.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("$tag", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"))
PY
done
for tag in base hist2; do
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  env $E PMC_TIMEOUT=200 timeout 300 python tools/pmc_kernels.py $OUT/pmc_$tag.json "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" -- python bench.py $B --steps 1 --warmup 1 > $OUT/pmc_$tag.log 2>&1
  echo "$tag $(grep kc_zstd_entropy_kernel $OUT/pmc_$tag.log | cut -c1-400)" | tee -a $OUT/summary.txt
done
