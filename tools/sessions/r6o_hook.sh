#!/bin/bash
# Session r6o: hook host-first rule with per-thread bookings: test + hook_bench.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6o
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/pytest_hook timeout 600 python -m pytest tests/test_gpu_s2.py -m gpu -q -x -k "custom_encoder"; echo "pytest rc $? $(tail -1 $OUT/pytest_hook.log)" | tee $OUT/summary.txt
for i in 1 2; do timeout 300 tools/_build/hook_bench oracle/_ref/libs2ref.so 2> $OUT/hook_bench.err | tee -a $OUT/hook_bench.jsonl | tee -a $OUT/summary.txt; done
