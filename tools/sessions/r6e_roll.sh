#!/bin/bash
# Session r6e: rolling host pipeline, 8 lanes on 4 queues + one-wave size scan: parity tests, rates at 2 and 3 calls in flight, timeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6e
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_s2.py -m gpu -q -x -k "rolling or chunk_fed or submit_wait or host_pipeline or serial or corpus_units or stream_framing" > $OUT/pytest_roll.log 2>&1; echo "pytest rc $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_roll.log | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --trace --steps 8 > $OUT/e2e_2.jsonl 2> $OUT/e2e_2.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/e2e_2.jsonl | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 --ctx 3 --steps 9 --only-two > $OUT/e2e_3.jsonl 2> $OUT/e2e_3.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/e2e_3.jsonl | tee -a $OUT/summary.txt
for c in C2 C4; do
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$c -- python tools/e2e_probe.py $c --only-two --steps 6 --ctx 3 > $OUT/e2e_$c.jsonl 2> $OUT/e2e_$c.err
tail -1 $OUT/e2e_$c.jsonl | cut -c1-600
python tools/trace_timeline.py $OUT/tr_$c 0.3 > $OUT/timeline_$c.txt 2>&1
rm -rf $OUT/tr_$c
done
