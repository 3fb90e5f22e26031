#!/bin/bash
# Session r7o: rolling pipeline, one call alone: graded first sub-batches (default) vs even cut (KC_HOST_ROLL_GRADED=0); C2 and C4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7o
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 600 python -m pytest tests/test_abi.py tests/test_zz_gpu_threads.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $? $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
for cfg in C2 C4; do
for rep in 1 2; do
for g in 0 1; do
  KC_HOST_ROLL_GRADED=$g timeout 400 python bench.py --config $cfg --no-also --no-cpu-baseline --no-floor --no-device-verify --steps 3 --warmup 1 2>$OUT/err_${cfg}_$g.txt | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); e=j['end_to_end']
print('$cfg graded=$g | device', j['value'], 'MB/s', j['ms_per_step'], 'ms | steady', e['value'], e['ms_per_batch'], 'ms', e['frac_of_device_resident'], '| single', e['single_call']['value'], e['single_call']['ms'], 'ms', e['single_call']['frac_of_device_resident'], 'same', e['same_bytes_as_device_path'])" | tee -a $OUT/summary.txt
done; done; done
KC_HOST_TRACE=1 timeout 400 python bench.py --config C2 --no-also --no-cpu-baseline --no-floor --no-device-verify --steps 2 --warmup 1 --e2e-calls 1 2>&1 >/dev/null | grep "kc roll" | tail -40 > $OUT/trace_C2_graded.txt
