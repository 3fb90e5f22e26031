#!/bin/bash
# Session r8j: bench.py --split (a step's batch as two launches of half the units, three contexts, two match finders in flight; frames
# contiguous through kc_zstd_encode_units_dev_end_at) against the arrangement so far (one launch per step, two contexts), same box,
# alternating; the new GPU test; one full line of the new default with the CPU checks.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8j}
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_zstd.py -q -x -k "one_batch_as_parts or begin_end_pipeline" 2>&1 | tail -3 | tee -a $OUT/summary.txt
one() {  # label, flags
  lab=$1; shift
  timeout 500 python bench.py --no-also --no-cpu-baseline --no-device-verify --no-end-to-end "$@" 2>$OUT/run.err | tail -1 > $OUT/run.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/run.json").read().strip().splitlines()[-1]); r = j["roofline"]; f = r.get("floor") or {}
    print("$lab |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "frac", r.get("frac"), r.get("frac_launches_in_flight"), "traffic", r.get("traffic"), "floor", f.get("floor_ms"), f.get("frac_of_floor"), "ctx", j.get("contexts"), "split", j.get("split"), "spread", j.get("ms_per_step_spread"))
except Exception as ex:
    print("$lab FAILED", ex, open("$OUT/run.err").read()[-400:])
PY
}
for rep in 1 2 3; do
  one "C2 before (2 ctx, 1 launch)" --config C2 --contexts 2 --split 1 --mf-in-flight 1 --steps 12 --warmup 3
  one "C2 default (3 ctx, 2 launches, 2 mf)" --config C2 --steps 12 --warmup 3
  one "C3 before (2 ctx, 1 launch)" --config C3 --contexts 2 --split 1 --mf-in-flight 1 --steps 6 --warmup 2
  one "C3 default (3 ctx, 2 launches, 2 mf)" --config C3 --steps 6 --warmup 2
done
one "C2 3 launches / 4 ctx / 3 mf" --config C2 --contexts 4 --split 3 --mf-in-flight 3 --steps 12 --warmup 3
one "C2 2 launches / 4 ctx / 2 mf" --config C2 --contexts 4 --split 2 --mf-in-flight 2 --steps 12 --warmup 3
timeout 900 python bench.py --no-also --steps 10 --warmup 3 2>$OUT/full.err | tail -1 > $OUT/full_C2.json
python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/full_C2.json").read().strip().splitlines()[-1]); e = j.get("end_to_end") or {}; c = j.get("cpu_baseline") or {}
print("C2 full line:", j["value"], j["ms_per_step"], "parity", j["bit_exact_vs_oracle_on_sample"], j["device_roundtrip_all_frames"], "reftr", (c.get("reference_translated") or {}).get("device_bytes_equal"), (c.get("reference_translated_parallel") or {}).get("device_bytes_equal"), "e2e", e.get("value"), e.get("frac_of_device_resident"), "ratio", j["ratio"])
PY
timeout 900 python bench.py --config C3 --no-also --no-end-to-end --steps 6 --warmup 2 2>$OUT/full.err | tail -1 > $OUT/full_C3.json
python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/full_C3.json").read().strip().splitlines()[-1]); c = j.get("cpu_baseline") or {}
print("C3 full line:", j["value"], j["ms_per_step"], "parity", j["bit_exact_vs_oracle_on_sample"], j["device_roundtrip_all_frames"], "reftr", (c.get("reference_translated") or {}).get("device_bytes_equal"), "ratio", j["ratio"])
PY
