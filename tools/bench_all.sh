#!/bin/bash
# One bench line per BASELINE.json configuration (1 GPU), written under gpurun_out/bench_<tag>/ on the GPU box.
TAG=${1:-r02}; STEPS=${2:-5}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
for c in C2 C2H C3 C4 C5; do
  python bench.py --config $c --steps $STEPS --warmup 1 > $OUT/$c.json 2> $OUT/$c.err
  tail -c 2200 $OUT/$c.json | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%s: %.0f MB/s, %.1f ms/step (median %.1f), ratio %.4f, kernel %.1f ms (frac %.5f), cpu %s, e2e %s, parity %s, roundtrip %s' % (
        j['config']['name'], j['value'], j['ms_per_step'], j['ms_per_step_median'], j['ratio'], j['roofline']['kernel_ms'], j['roofline']['frac'],
        j['cpu_baseline'] and j['cpu_baseline']['value'], j['end_to_end'] and j['end_to_end']['value'], j['bit_exact_vs_oracle_on_sample'], j['device_roundtrip_all_frames']))
except Exception as e:
    print('$c: FAILED', e); print(open('$OUT/$c.err').read()[-1500:])
"
done
