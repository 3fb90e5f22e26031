#!/bin/bash
# Session r6j: copy probe (layouts of the checksum-and-copy pass) + default bench after the trim fix.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6j
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 300 tools/_build/copy_probe > $OUT/copy_probe.txt 2>&1
cat $OUT/copy_probe.txt
( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc $?" | tee $OUT/summary.txt
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
also=d.pop('also',{})
e=d['end_to_end']; print('C2', d['value'], d['ms_per_step'], 'e2e', e.get('value'), e.get('frac_of_device_resident'), e.get('ms_per_batch'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'))
for k,v in also.items():
    e=v.get('end_to_end') or {}
    print(k, v.get('value'), v.get('ms_per_step'), 'e2e', e.get('value'), e.get('frac_of_device_resident'), e.get('frac_of_min_device_pcie'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'), 'floor', ((v.get('roofline') or {}).get('floor') or {}).get('frac_of_floor'))
" | tee -a $OUT/summary.txt
