#!/bin/bash
# Round 5, closing campaign on the MI355X: the whole GPU suite, smoke, the default bench line (all also-lines), kernel-trace stats of the
# configurations, and the counter traffic of every kernel whose source changed this round plus the N3 levels and the C2H pipeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5z
mkdir -p $OUT
cd $R
ulimit -c 0
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; grep -c "smoke ok" $OUT/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-400
fi
if [ -z "$SKIP_STATS" ]; then
for spec in "C2:--no-pipeline" "C2two:--pipeline" "C2H:" "C3:" "C5:" "C4:"; do
  name=${spec%%:*}; extra=${spec#*:}; cfg=${name%two}
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt --output-format csv -- python bench.py --config $cfg --steps 4 --warmup 1 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify $extra > $OUT/kt_$name.log 2>&1
  f=$(find $OUT/kt_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$name.csv && head -4 $f | cut -c1-150
  rm -rf $OUT/kt_$name
done
fi
if [ -z "$SKIP_PMC" ]; then
timeout 1500 python tools/pmc_update.py $OUT/pmc_traffic.json C2 C3 "B4 --gib 0.25" "C4 --s2-level 1 --gib 1.0" "C4 --s2-level 4 --gib 1.0" C2H 2>&1 | tee $OUT/pmc_update.log
fi
