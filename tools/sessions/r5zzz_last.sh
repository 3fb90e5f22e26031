#!/bin/bash
# Round 5, the very last call (the entropy kernel changed after the closing campaign: histogram copies, pack re-use): the whole GPU suite, smoke,
# the default bench line, kernel stats of C2 (one and two contexts) and C3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5zzz
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; grep -c "smoke ok" $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
for spec in "C2:--no-pipeline" "C2two:--pipeline" "C3:"; do
  name=${spec%%:*}; extra=${spec#*:}; cfg=${name%two}
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt --output-format csv -- python bench.py --config $cfg --steps 4 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify $extra > $OUT/kt_$name.log 2>&1
  f=$(find $OUT/kt_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$name.csv && head -3 $f | cut -c1-150
  rm -rf $OUT/kt_$name
done
