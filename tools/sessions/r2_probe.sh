#!/bin/bash
# Round-2 memory-system probe on the GPU box: microbench + PMC request-size counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2_probe
mkdir -p $OUT
cd $R
timeout 300 tools/_build/mem_probe 4096 512 > $OUT/mem_probe.txt 2>&1
cat $OUT/mem_probe.txt
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$tag -- tools/_build/mem_probe 4096 128 > $OUT/pmc_$tag.log 2>&1
  f=$(find $OUT/pmc_$tag -name "*.db" | head -1)
  python - "$f" <<'PY' > $OUT/pmc_$tag.txt 2>&1
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = "counters_collection"
cols = [x[1] for x in c.execute("pragma table_info(%s)" % t)]
ik, ic, iv, idp = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
acc = {}
for r in c.execute("select * from %s" % t):
    acc.setdefault((r[idp], r[ik][:60]), {}).setdefault(r[ic], 0.0)
    acc[(r[idp], r[ik][:60])][r[ic]] += float(r[iv])
for k in sorted(acc):
    print(k[0], k[1], " ".join("%s=%.4g" % kv for kv in sorted(acc[k].items())))
PY
  cat $OUT/pmc_$tag.txt | head -120
done
find $OUT -name "*.db" -delete
