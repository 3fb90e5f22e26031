#!/bin/bash
# Session r6c: the rolling host pipeline (kc_roll.cpp) on the device: its parity tests, then the host-buffer rates of C2 / C4 / C3 / C5
# (one call, two alternating contexts) against round 5's chunk-fed path (KC_HOST_ROLL=0) on the same box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6c
mkdir -p $OUT
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_s2.py -m gpu -q -x -k "rolling or chunk_fed or submit_wait or host_pipeline or serial" > $OUT/pytest_roll.log 2>&1; echo "pytest rc $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_roll.log | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --trace > $OUT/e2e_roll.jsonl 2> $OUT/e2e_roll.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cut -c1-600 $OUT/e2e_roll.jsonl | tee -a $OUT/summary.txt
KC_HOST_ROLL=0 timeout 600 python tools/e2e_probe.py C2 C4 > $OUT/e2e_chunkfed.jsonl 2> $OUT/e2e_chunkfed.err
cut -c1-600 $OUT/e2e_chunkfed.jsonl | tee -a $OUT/summary.txt
