#!/bin/bash
# Round 4: SQ counters of one S2 LDS-kernel launch (one 64 KiB block): where a lone wave's cycles go.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4g
cd $R
python - <<PY
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import corpora
open("/tmp/J.bin", "wb").write(corpora.corpus("J", 1, 65536).tobytes())
open("/tmp/T.bin", "wb").write(corpora.corpus("T", 1, 65536).tobytes())
PY
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -I compress_amd/csrc tools/s2_lds_prof.hip"
/opt/rocm/bin/hipcc $F -o /tmp/s2noprof 2>/dev/null
for mode in ${MODES:-0 1}; do
for k in J; do
  rm -rf /tmp/pmc_$k
  timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d /tmp/pmc_$k -o out --output-format csv -- /tmp/s2noprof /tmp/$k.bin $mode > /tmp/pmc_$k.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$k/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f[0])):
    if "kc_s2_encode_lds" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        n += 1
print("$k mode $mode (3 launches summed):", {k: v for k, v in acc.items()})
PY
done
done 2>&1 | tee $R/gpurun_out/r4g/pmc.txt
