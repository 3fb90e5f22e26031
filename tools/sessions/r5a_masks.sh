#!/bin/bash
# Session r5a (GPU box, repo root): (1) the CU-mask probe — match finder on 256 / 224 / 208 / 192 / 128 CUs, the entropy stage on the
# complement (KC_OPT_STAGE2_STREAM), product library and the 512-byte-ring measurement build (the match finder leaves LDS for a
# co-resident entropy workgroup); (2) DRAM requests of the entropy kernel beside the match finder's (one context, C2): is the
# co-resident entropy stage paid in DRAM transactions?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5a
mkdir -p $OUT
cd $R
timeout 400 python tools/cu_mask_probe.py $OUT/cu_mask.json > $OUT/cu_mask.log 2>&1; echo "probe rc=$? $(date +%T)"
PROBE_PARTS=2 KC_LIB_TAG=rb512 timeout 300 python tools/cu_mask_probe.py $OUT/cu_mask_rb512.json > $OUT/cu_mask_rb512.log 2>&1; echo "probe rb512 rc=$? $(date +%T)"
ARGS="bench.py --config C2 --steps 1 --warmup 1 --no-cpu-baseline --no-device-verify --no-end-to-end --no-also --no-pipeline"
PMC_TIMEOUT=200 timeout 900 python tools/pmc_kernels.py $OUT/pmc_C2_dram.json \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "FETCH_SIZE" "WRITE_SIZE" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  -- python $ARGS > $OUT/pmc_C2_dram.log 2>&1; echo "pmc rc=$? $(date +%T)"
tail -5 $OUT/cu_mask.log; tail -3 $OUT/cu_mask_rb512.log; tail -12 $OUT/pmc_C2_dram.log | cut -c1-600
