#!/bin/bash
# Session r6g: kernel timeline of C5 (SpeedBetter + dictionary) with two calls in flight through the rolling pipeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6g
mkdir -p $OUT
cd $R
ulimit -c 0
for c in C5; do
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$c -- python tools/e2e_probe.py $c --only-two --steps 6 --ctx 2 > $OUT/e2e_$c.jsonl 2> $OUT/e2e_$c.err
tail -1 $OUT/e2e_$c.jsonl | cut -c1-600
python tools/trace_timeline.py $OUT/tr_$c 0.3 > $OUT/timeline_$c.txt 2>&1
rm -rf $OUT/tr_$c
done
