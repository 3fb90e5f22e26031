#!/bin/bash
# PMC counters of the dominant kernel on the headline launch: one rocprofv3 pass per counter group (--pmc only with --kernel-trace).
# Usage (repo root on the GPU box): bash tools/pmc_match.sh <tag> [bench args...]
TAG=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $R
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-device-verify --no-end-to-end "$@" > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*.db" | head -1)
  python tools/pmc_summary.py $f 2>&1 | grep -E "pmc.*(match|s2_encode)" | tee -a $OUT/summary.txt
done
find $OUT -name "*.db" -delete
