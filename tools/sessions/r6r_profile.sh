#!/bin/bash
# Session r6r: the round's profile (tools/profile_round.sh r06: bench lines, rocprofv3 kernel-trace stats per configuration, HBM traffic
# of the kernels whose sources changed) + DRAM request counts of C4 / C5 for roofline.floor.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
PMC_CONFIGS="C4 C4A C5" bash tools/profile_round.sh r06 > gpurun_out/profile_r06.log 2>&1
tail -12 gpurun_out/profile_r06.log | cut -c1-300
OUT=$R/gpurun_out/prof_r06
for c in C4 C5; do
  B="--config $c --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --no-floor --steps 1 --warmup 1 --no-pipeline"
  PMC_TIMEOUT=200 timeout 300 python tools/pmc_kernels.py $OUT/tx_$c.json "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- python bench.py $B > $OUT/tx_$c.log 2>&1
  grep -E "kc_(zbetter|s2_encode)_" $OUT/tx_$c.log | cut -c1-300
done
