#!/bin/bash
# Session r5b (GPU box, repo root): the entropy stage with register-resident tANS chain segments and the per-wave literal gather against
# round 4's kernel (KC_LIB_TAG=oldent: HEAD~'s kc_zstd_entropy.hip in today's library), the chains' warm-up at 48 / 32 / 16 symbols; GPU
# parity subset first.  One context, so that entropy_kernel_ms is the kernel alone; then the two-context line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5b
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_zstd.py -x -q -m gpu -k "corpus_units or edge or stress or ragged or raw_only or rle_literal or long_units" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_subset.log)"
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --steps 5 --warmup 2"
for tag in oldent base warm32 warm16 oldent base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 200 python bench.py $B --no-pipeline > $OUT/${tag}.json 2> $OUT/${tag}.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "roundtrip", j.get("device_roundtrip_all_frames"))
except Exception as e:
    print("$tag FAILED", e, open("$OUT/${tag}.err").read()[-300:])
PY
done 2>&1 | tee $OUT/summary.txt
for tag in oldent base; do
    E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    env $E timeout 200 python bench.py $B --no-device-verify --pipeline --steps 8 > $OUT/${tag}_p1.json 2> $OUT/${tag}_p1.err
    python - <<PY
import json
j = json.loads(open("$OUT/${tag}_p1.json").read().strip().splitlines()[-1]); r = j["roofline"]
print("$tag two contexts", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"))
PY
done 2>&1 | tee -a $OUT/summary.txt
KC_K2_PROF=1 timeout 150 python bench.py $B --no-pipeline --no-device-verify --steps 1 --warmup 1 > $OUT/k2prof.json 2> $OUT/k2prof.err
grep "K2 prof" $OUT/k2prof.err | tail -2 | tee -a $OUT/summary.txt
PMC_TIMEOUT=200 timeout 500 python tools/pmc_kernels.py $OUT/pmc_entropy.json \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
  "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
  -- python bench.py $B --no-pipeline --no-device-verify --steps 1 --warmup 1 > $OUT/pmc_entropy.log 2>&1
grep "kc_zstd_entropy_kernel" $OUT/pmc_entropy.log | cut -c1-900 | tee -a $OUT/summary.txt
