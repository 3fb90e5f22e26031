#!/bin/bash
# Session r4u (GPU box, repo root): why kc_zstd_entropy_kernel stays at 20 ms per 4 GiB (C2) — SQ counters of its dispatch on the
# headline launch (instructions by kind, busy / wait cycles, LDS bank conflicts) and the per-phase shader-clock shares (KC_K2_PROF).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4u
mkdir -p $OUT
cd $R
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-device-verify --no-end-to-end --no-also --no-pipeline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmce
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmce -o out --output-format csv -- python bench.py $ARGS > $OUT/p$i.log 2>&1
  python - <<PY | tee -a $OUT/entropy_pmc.txt
import csv, glob, collections
f = glob.glob("/tmp/pmce/**/*counter_collection.csv", recursive=True)
if not f:
    print("pass $i: no counter file (a counter name this chip does not have?)")
else:
    for key in ("kc_zstd_entropy_kernel", "kc_zfast_match_grp_kernel"):
        acc = collections.defaultdict(float); n = 0; disp = set()
        for r in csv.DictReader(open(f[0])):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r.get("Dispatch_Id"))
        print("pass $i", key, "dispatches", len(disp), {k: v for k, v in sorted(acc.items())})
PY
done
KC_K2_PROF=1 timeout 150 python bench.py $ARGS > $OUT/k2prof.json 2> $OUT/k2prof.err
grep "K2 prof" $OUT/k2prof.err | tail -4 | tee -a $OUT/entropy_pmc.txt
tail -c 600 $OUT/k2prof.json | tr ',' '\n' | grep -E "entropy_kernel_ms|kernel_ms|ms_per_step\"" | tee -a $OUT/entropy_pmc.txt
