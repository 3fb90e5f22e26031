#!/bin/bash
# Round 5, last call: the device against the reference's own Go encoder (translated) on random units — EncodeAll at the four levels, streams,
# dictionaries (tools/fuzz_zstd_goref.py, several seeds) — and the default bench line with the refreshed profiles/pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5zz
mkdir -p $OUT
cd $R
ulimit -c 0
for seed in 51 52 53; do
  timeout 400 python tools/fuzz_zstd_goref.py 400 $seed 2>&1 | grep -E "^level" | sed "s/^/seed $seed: /" | tee -a $OUT/fuzz_zstd_device.txt
done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
