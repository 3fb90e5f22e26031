#!/bin/bash
# Session r6d: rolling host pipeline with the lanes on the low-priority queue pool: rates + the kernel timeline of C2's steady state.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6d
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --trace > $OUT/e2e_roll.jsonl 2> $OUT/e2e_roll.err; echo "e2e rc $?" | tee $OUT/summary.txt
cut -c1-600 $OUT/e2e_roll.jsonl | tee -a $OUT/summary.txt
for c in C2 C4; do
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$c -- python tools/e2e_probe.py $c --only-two --steps 4 > $OUT/e2e_$c.jsonl 2> $OUT/e2e_$c.err
tail -1 $OUT/e2e_$c.jsonl | cut -c1-600
python tools/trace_timeline.py $OUT/tr_$c 0.3 > $OUT/timeline_$c.txt 2>&1
rm -rf $OUT/tr_$c
done
