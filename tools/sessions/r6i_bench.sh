#!/bin/bash
# Session r6i: the default bench line (floor, end_to_end through the rolling pipeline, also-lines with end_to_end) + the whole GPU suite.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6i
mkdir -p $OUT
cd $R
ulimit -c 0
( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc $?" | tee $OUT/summary.txt
tail -3 $OUT/bench_time.txt | tee -a $OUT/summary.txt
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
also=d.pop('also',{})
print(json.dumps({k:d[k] for k in ('value','ms_per_step','roofline','end_to_end','bit_exact_vs_oracle_on_sample','device_roundtrip_all_frames')})[:3500])
for k,v in also.items(): print(k, json.dumps({x:v.get(x) for x in ('value','ms_per_step','end_to_end','error')})[:900]); print('   floor', json.dumps((v.get('roofline') or {}).get('floor'))[:600])
" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
