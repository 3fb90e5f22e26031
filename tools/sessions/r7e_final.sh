#!/bin/bash
# Session r7e: closing state of round 6 after the best-level work: HBM traffic of the side lines at their new sizes, then the driver's
# three commands (smoke, default bench, pytest -m gpu).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r7e
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 1500 python tools/pmc_update.py $OUT/pmc_traffic.json B4 "C4 --s2-level 1" "C4 --s2-level 4 --gib 1.5" 2>&1 | tail -5 | tee $OUT/summary.txt
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
    print(k, v.get('value'), v.get('ms_per_step'), 'ctx', v.get('contexts'), 'traffic', r.get('traffic'), 'e2e', e.get('value'), e.get('frac_of_device_resident'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'), v.get('error'), 'floor', (r.get('floor') or {}).get('frac_of_floor'), 'parity', v.get('bit_exact_vs_oracle_on_sample'), v.get('device_roundtrip_all_frames'))
" | tee -a $OUT/summary.txt
bash tools/gpu_guard.sh $OUT/pytest_gpu timeout 1800 python -m pytest tests -m gpu -q -x; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu.log)" | tee -a $OUT/summary.txt
