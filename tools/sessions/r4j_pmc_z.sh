#!/bin/bash
# Round 4: SQ counters of the SpeedFastest LDS kernels on ONE 128 KiB unit of text (one wave): instructions and wait cycles per kernel form.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4j
cd $R
cat > /tmp/one.py <<PY
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, corpora
from compress_amd import zstd
mode = int(sys.argv[1])
usz = 131072
buf = corpora.corpus("T", 1, usz)
d_src = torch.from_numpy(buf).cuda()
enc = zstd.NewWriter(None, zstd.WithEncoderLevel(1), zstd.WithMatchPath("lds"))
enc.ctx().set_option(6, mode)
cap = ((enc.MaxEncodedSize(usz) + 15) & ~15) + 64
d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
off = np.arange(2, dtype=np.uint64) * usz
for _ in range(3):
    enc.EncodeUnitsDevice(d_src.data_ptr(), off, d_dst.data_ptr(), cap)
torch.cuda.synchronize()
PY
for mode in ${MODES:-16 0 -1}; do
  rm -rf /tmp/pmcz
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d /tmp/pmcz -o out --output-format csv -- python /tmp/one.py $mode > /tmp/pmcz.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmcz/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "kc_zfast_match_lds" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]] += 1
print("mode $mode (3 launches summed):", dict(acc), dict(n))
PY
done 2>&1 | tee $R/gpurun_out/r4j/pmc_z.txt
