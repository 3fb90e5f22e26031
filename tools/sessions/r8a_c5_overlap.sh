#!/bin/bash
# Session r8a: bench.py chains the match finders of its contexts (one at a time: right for C2, whose kernel is one residency of the chip).  C5's is not:  (150 VGPRs = 3 waves per SIMD = 12 per CU; one
# batch is 2 048 one-wave workgroups = 8 per CU.)  Part 1: kernel-trace timeline of the default arrangement (three contexts).  Part 2: the
# kernel held to 128 VGPRs (zbw4: amdgpu_waves_per_eu(4,4), 84 B of scratch: two batches' kernels fit the CUs together) against the
# product, same box, alternating, at 1 / 3 / 4 / 5 contexts.  Part 3: the new cpu_baseline.reference_translated_parallel leg.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${SESSION:-r8a}
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C5 --no-also --no-cpu-baseline --no-end-to-end --no-floor --no-device-verify"
for tag in base zbw4; do
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  env $E timeout 600 rocprofv3 --kernel-trace -d $OUT/kt_$tag -- python bench.py $B --contexts 4 --mf-in-flight 2 --steps 8 --warmup 4 > $OUT/kt_$tag.log 2>&1
  python - <<PY | tee $OUT/timeline_$tag.txt
import sqlite3, glob, os
f = glob.glob(os.path.join("$OUT", "kt_$tag", "**", "*.db"), recursive=True)
k = sqlite3.connect(f[0])
tabs = [r[0] for r in k.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if t == "kernels"]
rows = list(k.execute("select name, start, end, queue_id, stream_id from kernels order by start")) if v else []
if not rows: print("no kernels view; tables:", tabs[:40])
else:
    t0 = rows[0][1]
    big = [r for r in rows if (r[2] - r[1]) > 300000]
    # the last 40 kernels longer than 0.3 ms: steady state
    for r in big[-40:]:
        print("%-34s %9.2f -> %9.2f  (%6.2f ms)  queue %s stream %s" % (r[0][:34], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], r[4]))
    mf = [r for r in rows if "zbetter" in r[0]]
    ov = 0.0
    for a, b in zip(mf[-8:], mf[-7:]):
        ov += max(0, min(a[2], b[2]) - max(a[1], b[1]))
    print("match finders: last 8 durations", [round((r[2] - r[1]) / 1e6, 1) for r in mf[-8:]], "pairwise overlap of neighbours, ms:", round(ov / 1e6, 1))
PY
  find $OUT/kt_$tag -name "*.db" -delete; rm -rf $OUT/kt_$tag
done
for rep in 1 2; do
for tag in base zbw4; do
  E="KC_X=0"; [ $tag != base ] && E="KC_LIB_TAG=$tag"
  for mode in "--no-pipeline --steps 8 --warmup 3" "--steps 12 --warmup 6" "--contexts 3 --mf-in-flight 2 --steps 12 --warmup 6" "--contexts 4 --mf-in-flight 2 --steps 12 --warmup 8" "--contexts 4 --mf-in-flight 3 --steps 12 --warmup 8"; do
    env $E timeout 300 python bench.py $B $mode 2>$OUT/$tag.err | tail -1 > $OUT/$tag.json
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    j = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("$tag $mode |", j["value"], "MB/s", j["ms_per_step"], "ms/step; kernel", r.get("kernel_ms"), "entropy", r.get("entropy_kernel_ms"), "ctx", j.get("contexts"))
except Exception as e:
    print("$tag $mode FAILED", e, open("$OUT/$tag.err").read()[-300:])
PY
  done
done
done
timeout 600 python bench.py --no-also --no-end-to-end --steps 5 --warmup 2 2>$OUT/c2.err | tail -1 > $OUT/c2.json
python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/c2.json").read().strip().splitlines()[-1]); c = j["cpu_baseline"]
print("C2", j["value"], "cpu", c.get("value"), c.get("cores"), "reftr", c.get("reference_translated"), "\nreftr parallel", c.get("reference_translated_parallel"))
PY
timeout 600 python bench.py --config C5 --no-also --no-end-to-end --steps 5 --warmup 3 2>$OUT/c5.err | tail -1 > $OUT/c5.json
python - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads(open("$OUT/c5.json").read().strip().splitlines()[-1]); c = j["cpu_baseline"]
print("C5", j["value"], "cpu", c.get("value"), c.get("cores"), "reftr parallel", c.get("reference_translated_parallel"))
PY
