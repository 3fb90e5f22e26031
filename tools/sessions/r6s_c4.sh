#!/bin/bash
# Session r6s: kernel + copy timeline of C4's steady state through the rolling pipeline (final code, 4 calls in flight).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6s
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_C4 -- python tools/e2e_probe.py C4 --only-two --steps 8 --ctx 4 > $OUT/e2e_C4.jsonl 2> $OUT/e2e_C4.err
tail -1 $OUT/e2e_C4.jsonl | cut -c1-600
python tools/trace_timeline.py $OUT/tr_C4 0.2 > $OUT/timeline_C4.txt 2>&1
rm -rf $OUT/tr_C4
