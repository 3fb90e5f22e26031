#!/bin/bash
# Session r6n: hook (host-first rule: test + tools/hook_bench against the reference's assembly encoder as the built-in), trim test,
# device self-check, one serialized run of a parity subset (AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3), smoke.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6n
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/pytest_new timeout 600 python -m pytest tests/test_gpu_s2.py tests/test_gpu_zstd.py tests/test_abi.py -m gpu -q -x -k "custom_encoder or trim or rolling or abi or option"; echo "pytest new rc $? $(tail -1 $OUT/pytest_new.log)" | tee $OUT/summary.txt
timeout 300 tools/_build/hook_bench oracle/_ref/libs2ref.so > $OUT/hook_bench.json 2> $OUT/hook_bench.err; echo "hook_bench rc $?" | tee -a $OUT/summary.txt
cat $OUT/hook_bench.json | tee -a $OUT/summary.txt
AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 bash tools/gpu_guard.sh $OUT/pytest_serialized timeout 900 python -m pytest tests/test_gpu_zstd.py -m gpu -q -x -k "corpus_units or rolling or chunk_fed or begin_end"; echo "serialized rc $? $(tail -1 $OUT/pytest_serialized.log)" | tee -a $OUT/summary.txt
bash tools/gpu_guard.sh $OUT/smoke timeout 300 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc $?" | tee -a $OUT/summary.txt
tail -8 $OUT/smoke.log | tee -a $OUT/summary.txt
