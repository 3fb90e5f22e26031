#!/bin/bash
# Session r6f: rolling host pipeline with queue-aware lane assignment + S2 arena pre-clear: parity tests, rates at 1 / 2 / 3 calls in flight.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6f
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_s2.py -m gpu -q -x -k "rolling or chunk_fed or submit_wait or host_pipeline or serial or corpus_units or stream_framing or budget" > $OUT/pytest_roll.log 2>&1; echo "pytest rc $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_roll.log | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --trace --steps 8 > $OUT/e2e_2.jsonl 2> $OUT/e2e_2.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/e2e_2.jsonl | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --ctx 3 --steps 9 --only-two > $OUT/e2e_3.jsonl 2> $OUT/e2e_3.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/e2e_3.jsonl | tee -a $OUT/summary.txt
