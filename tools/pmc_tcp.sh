cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
cd $R
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_WRITE_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc2/p$i -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc2/p$i.log 2>&1
  f=$(find gpurun_out/pmc2/p$i -name "*.db" | head -1)
  echo "== $set"
  python tools/pmc_summary.py $f 2>&1 | grep -E "pmc.*match_grp" 
done
find gpurun_out/pmc2 -name "*.db" -delete
