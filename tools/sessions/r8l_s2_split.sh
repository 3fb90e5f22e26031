#!/bin/bash
# Session r8l: s2.Encode's batch as two launches on three contexts (kc_s2_encode_blocks_lvl_dev_begin / _end_at; bench.py's new default for
# C4 / C4A and the better level) against one launch per step, same box, alternating; the new GPU tests.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8l}
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_s2.py tests/test_gpu_zstd.py -q -x -k "one_batch_as_parts" 2>&1 | tail -3 | tee -a $OUT/summary.txt
one() {  # label, flags
  lab=$1; shift
  timeout 500 python bench.py --no-also --no-cpu-baseline --no-device-verify --no-end-to-end "$@" 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); r = j["roofline"]; f = r.get("floor") or {}
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "frac", r.get("frac"), r.get("frac_launches_in_flight"), r.get("step_frac"), "traffic", r.get("traffic"), "floor", f.get("floor_ms"), f.get("frac_of_floor"), "ctx", j.get("contexts"), "split", j.get("split"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-400:])
PY
}
for rep in 1 2 3; do
  one "C4 one launch per step" --config C4 --no-pipeline --steps 20 --warmup 4
  one "C4 default (3 ctx, 2 launches)" --config C4 --steps 20 --warmup 4
done
one "C4 4 ctx, 2 launches" --config C4 --contexts 4 --split 2 --steps 20 --warmup 4
one "C4 2 ctx, 2 launches" --config C4 --contexts 2 --split 2 --steps 20 --warmup 4
one "C4 3 ctx, 3 launches" --config C4 --contexts 3 --split 3 --steps 20 --warmup 4
one "C4A one launch per step" --config C4A --no-pipeline --steps 20 --warmup 4
one "C4A default" --config C4A --steps 20 --warmup 4
one "s2.EncodeBetter one launch per step" --config C4 --s2-level 1 --no-pipeline --steps 8 --warmup 2
one "s2.EncodeBetter default" --config C4 --s2-level 1 --steps 8 --warmup 2
one "s2.EncodeSnappy default" --config C4 --s2-level 2 --steps 8 --warmup 2
timeout 900 python bench.py --config C4 --no-also --steps 10 --warmup 3 2>$OUT/full.err | tail -1 > $OUT/full_C4.json
python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/full_C4.json").read().strip().splitlines()[-1]); e = j.get("end_to_end") or {}; c = j.get("cpu_baseline") or {}
print("C4 full line:", j["value"], j["ms_per_step"], "parity", j["bit_exact_vs_oracle_on_sample"], j["device_roundtrip_all_frames"], "e2e", e.get("value"), e.get("frac_of_device_resident"), (e.get("single_call") or {}).get("value"), "ratio", j["ratio"], "cpu", c.get("value"))
PY
