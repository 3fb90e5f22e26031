#!/bin/bash
# Round 4: does the entropy stage of batch i hide under the match finder of batch i+1 (bench.py --pipeline: two contexts, two
# streams) once the match finder leaves LDS for it (ZW_RB=512: 5 KiB instead of 8.75 KiB per 8-unit workgroup)?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4b
mkdir -p $OUT
cd $R
B="--config C2 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 6 --warmup 2"
for tag in ${TAGS:-base rb512 rb512prio prio}; do
  for pipe in 0 1; do
    E=""; [ $tag != base ] && E="KC_LIB_TAG=$tag"
    P=""; [ $pipe = 1 ] && P="--pipeline"
    env $E timeout 200 python bench.py $B $P > $OUT/${tag}_p$pipe.json 2> $OUT/${tag}_p$pipe.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}_p$pipe.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag pipe=$pipe", j["value"], "MB/s", j["ms_per_step"], "ms/step (median", j.get("ms_per_step_median"), "); kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "sample ok", j.get("bit_exact_vs_oracle_on_sample"))
except Exception as e:
    print("$tag pipe=$pipe FAILED", e, open("$OUT/${tag}_p$pipe.err").read()[-400:])
PY
  done
done
