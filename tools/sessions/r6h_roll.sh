#!/bin/bash
# Session r6h: rolling host pipeline, warmed lanes: rates at 2 / 3 calls in flight for the four BASELINE configurations.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6h
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --steps 8 > $OUT/e2e_2.jsonl 2> $OUT/e2e_2.err; echo "e2e rc $?" | tee $OUT/summary.txt
cut -c1-600 $OUT/e2e_2.jsonl | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --ctx 3 --steps 9 --only-two > $OUT/e2e_3.jsonl 2> $OUT/e2e_3.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/e2e_3.jsonl | tee -a $OUT/summary.txt
