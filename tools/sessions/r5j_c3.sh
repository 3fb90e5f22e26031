#!/bin/bash
# Session r5j (GPU box, repo root): C3's match finder — what a unit asks of the memory system by source (KC_LIB_TAG=zdstats:
# -DKC_ZD_STATS counters) beside what the memory system sees (TCC_EA0_RDREQ / WRREQ, FETCH_SIZE, WRITE_SIZE per dispatch).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5j
mkdir -p $OUT
cd $R
ulimit -c 0
B="--config C3 --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 1 --warmup 1 --no-pipeline"
KC_LIB_TAG=zdstats KC_K2_PROF=1 timeout 200 python bench.py $B > $OUT/zdstats.json 2> $OUT/zdstats.err
grep "dfast stats" $OUT/zdstats.err | tail -1 | tee $OUT/summary.txt
PMC_TIMEOUT=200 timeout 700 python tools/pmc_kernels.py $OUT/pmc_C3.json "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" -- python bench.py $B > $OUT/pmc_C3.log 2>&1
grep "kc_zdfast" $OUT/pmc_C3.log | cut -c1-500 | tee -a $OUT/summary.txt
