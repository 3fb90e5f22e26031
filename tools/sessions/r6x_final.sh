#!/bin/bash
# Session r6x: the round's closing state — HBM traffic + DRAM requests of C5's kernel on its final source, then the driver's three
# commands (smoke, default bench, pytest -m gpu).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6x
mkdir -p $OUT
cd $R
ulimit -c 0
python bench.py --config C5 --steps 2 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --no-floor --no-pipeline --pmc > $OUT/pmc_C5.json 2> $OUT/pmc_C5.err
tail -1 $OUT/pmc_C5.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('C5 traffic', j['roofline']['traffic'], j['roofline']['traffic_source'], j['roofline']['kernel_ms'], j['ratio'])" | tee $OUT/summary.txt
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --no-floor --steps 1 --warmup 1 --no-pipeline"
PMC_TIMEOUT=200 timeout 300 python tools/pmc_kernels.py $OUT/tx_C5.json "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- python bench.py $B > $OUT/tx_C5.log 2>&1
grep -E "kc_zbetter" $OUT/tx_C5.log | cut -c1-300 | tee -a $OUT/summary.txt
bash tools/gpu_guard.sh $OUT/smoke timeout 400 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc $?" | tee -a $OUT/summary.txt
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc $? $(grep real $OUT/bench_time.txt)" | tee -a $OUT/summary.txt
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
also=d.pop('also',{})
e=d['end_to_end']; f=d['roofline'].get('floor') or {}
print('C2', d['value'], d['ms_per_step'], 'floor', f.get('floor_ms'), f.get('frac_of_floor'), 'e2e', e.get('value'), e.get('frac_of_device_resident'), e.get('ms_per_batch'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'), 'parity', d['bit_exact_vs_oracle_on_sample'], d['device_roundtrip_all_frames'])
for k,v in also.items():
    e=v.get('end_to_end') or {}; r=v.get('roofline') or {}
    print(k, v.get('value'), v.get('ms_per_step'), 'ctx', v.get('contexts'), 'traffic', r.get('traffic'), 'e2e', e.get('value'), e.get('frac_of_device_resident'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'), v.get('error'), 'floor', (r.get('floor') or {}).get('frac_of_floor'), (r.get('floor') or {}).get('transactions_source_current'), 'parity', v.get('bit_exact_vs_oracle_on_sample'), v.get('device_roundtrip_all_frames'))
" | tee -a $OUT/summary.txt
bash tools/gpu_guard.sh $OUT/pytest_gpu timeout 1800 python -m pytest tests -m gpu -q -x; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu.log)" | tee -a $OUT/summary.txt
