#!/bin/bash
# Session r6v: SpeedBetter with a dictionary: per-batch table copy (0, default) vs stamped tables + shared dictionary table (1, round 3)
# vs the same with the 64 KiB bucket map in front of the shared table (2, round 6), on C5; dictionary parity tests first (all modes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6v
mkdir -p $OUT
cd $R
ulimit -c 0
for m in 0 2; do
KC_BETTER_DICT_EPOCH=$m bash tools/gpu_guard.sh $OUT/pytest_dict_$m timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_ref_inputs.py -m gpu -q -x -k "dict or rolling"; echo "pytest mode $m rc $? $(tail -1 $OUT/pytest_dict_$m.log)" | tee -a $OUT/summary.txt
done
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 --no-pipeline --no-floor"
for m in 0 1 2 0 1 2; do
  KC_BETTER_DICT_EPOCH=$m timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('mode $m', j['value'], 'MB/s', j['ms_per_step'], 'ms/step; kernel', r['kernel_ms'], 'prep', r['table_prep_ms'], 'roundtrip', j['device_roundtrip_all_frames'], 'parity', j['bit_exact_vs_oracle_on_sample'])" | tee -a $OUT/summary.txt
done
