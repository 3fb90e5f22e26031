#!/bin/bash
# Session r6a (GPU box, repo root): where the host-buffer entry points lose against the device-resident rate — this box's PCIe and
# host-copy ceilings (kc_probe_pcie), one call and two alternating contexts with the chunk timeline (KC_OPT_HOST_TRACE) for C2 / C4 /
# C3 / C5; the table-pattern rates (kc_probe_table_pattern) and the match finders' DRAM requests per dispatch (TCC_EA0_RDREQ / WRREQ)
# that bench.py's roofline.floor is built from.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6a
mkdir -p $OUT
cd $R
ulimit -c 0
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt; lscpu | head -20 >> $OUT/host.txt; numactl -H >> $OUT/host.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" | tee -a $OUT/summary.txt
timeout 600 python tools/e2e_probe.py C2 C4 C3 C5 --trace > $OUT/e2e.jsonl 2> $OUT/e2e.err; echo "e2e rc $?" | tee -a $OUT/summary.txt
cat $OUT/e2e.jsonl | cut -c1-700 | tee -a $OUT/summary.txt
timeout 200 python - > $OUT/table_probe.json 2>> $OUT/summary.txt <<PY
import json, sys, os
sys.path.insert(0, os.getcwd())
from compress_amd import _lib
c = _lib.Context(0)
res = {}
for nt in (32768, 8192, 2048, 512):
    res[str(nt)] = c.probe_table_pattern(nt, 131072, 4096, 512)
res["640KiB_x_32768"] = c.probe_table_pattern(32768, 1 << 19, 4096, 512)
print(json.dumps(res, indent=1))
PY
cat $OUT/table_probe.json | tr -d '\n' | cut -c1-900 | tee -a $OUT/summary.txt; echo
for c in C2 C4 C5 C3; do
  B="--config $c --no-also --no-cpu-baseline --no-end-to-end --no-device-verify --steps 1 --warmup 1 --no-pipeline"
  PMC_TIMEOUT=200 timeout 300 python tools/pmc_kernels.py $OUT/tx_$c.json "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- python bench.py $B > $OUT/tx_$c.log 2>&1
  grep -E "kc_(zfast_match_grp|zdfast|zbetter|s2_encode)_kernel" $OUT/tx_$c.log | cut -c1-300 | tee -a $OUT/summary.txt
done
