#!/bin/bash
# Session r8r: closing state of round 6 (a batch as two launches: C2 / C3 / C4; two match finders in flight: C5): kernel-trace stats of the four BASELINE configurations under the bench defaults, then the driver's three commands (smoke, default bench, pytest -m gpu).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r8r
mkdir -p $OUT
cd $R
ulimit -c 0
for c in C5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$c -- python bench.py --config $c --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --no-floor > $OUT/kt_$c.log 2>&1
done
python - <<PY
import sqlite3, glob, csv, os
out = "$OUT"
for tag in ("C5",):
    f = glob.glob(os.path.join(out, "kt_" + tag, "**", "*.db"), recursive=True)
    if not f: print("no db for", tag); continue
    k = sqlite3.connect(f[0])
    rows = list(k.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(out, "kernel_stats_%s.csv" % tag), "w") as fo:
        w = csv.writer(fo); w.writerow(["name", "total_calls", "total_duration_ms", "average_ms", "percentage"])
        for r in rows: w.writerow([r[0][:120], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4], 3)])
    print(tag, [(r[0][:40], round(r[3] / 1e3, 3)) for r in rows[:3]])
    try:
        import json
        j = json.loads([l for l in open(os.path.join(out, "kt_%s.log" % tag)) if l.startswith("{")][-1]); r = j["roofline"]
        print(tag, "bench line of the same run: value", j["value"], "ms_per_step", j["ms_per_step"], "roofline.kernel_ms (mean of the timed launches, HIP events)", r["kernel_ms"], "launches per step", r.get("launches_per_step"), "in flight", r.get("match_finders_in_flight"))
    except Exception as e:
        print(tag, "no bench line", e)
PY
find $OUT -name "*.db" -delete; find $OUT -type d -name "kt_*" | xargs rm -rf
bash tools/gpu_guard.sh $OUT/smoke timeout 400 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc $?" | tee -a $OUT/summary.txt
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench rc $? $(grep real $OUT/bench_time.txt)" | tee -a $OUT/summary.txt
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
also=d.pop('also',{})
e=d['end_to_end']; f=d['roofline'].get('floor') or {}
print('C2', d['value'], d['ms_per_step'], 'floor', f.get('floor_ms'), f.get('frac_of_floor'), 'e2e', e.get('value'), e.get('frac_of_device_resident'), e.get('ms_per_batch'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'), 'parity', d['bit_exact_vs_oracle_on_sample'], d['device_roundtrip_all_frames'], 'reftr', ((d['cpu_baseline'] or {}).get('reference_translated') or {}).get('value'), 'reftr-parallel', (d['cpu_baseline'] or {}).get('reference_translated_parallel'))
for k,v in also.items():
    e=v.get('end_to_end') or {}; r=v.get('roofline') or {}
    print(k, v.get('value'), v.get('ms_per_step'), 'ctx', v.get('contexts'), 'traffic', r.get('traffic'), 'e2e', e.get('value'), e.get('frac_of_device_resident'), 'single', (e.get('single_call') or {}).get('value'), e.get('error'), v.get('error'), 'floor', (r.get('floor') or {}).get('frac_of_floor'), 'parity', v.get('bit_exact_vs_oracle_on_sample'), v.get('device_roundtrip_all_frames'), 'reftr', ((v.get('cpu_baseline') or {}).get('reference_translated') or {}).get('device_bytes_equal'), 'reftr-parallel', ((v.get('cpu_baseline') or {}).get('reference_translated_parallel') or {}).get('value'), ((v.get('cpu_baseline') or {}).get('reference_translated_parallel') or {}).get('device_bytes_equal'), 'inflight', r.get('match_finders_in_flight'))
" | tee -a $OUT/summary.txt
bash tools/gpu_guard.sh $OUT/pytest_gpu timeout 1800 python -m pytest tests -m gpu -q -x; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu.log)" | tee -a $OUT/summary.txt
