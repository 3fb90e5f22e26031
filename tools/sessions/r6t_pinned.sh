#!/bin/bash
# Session r6t: page-locked caller buffers (kc_host_alloc): parity test, then C2 / C4 rates with pageable vs pinned buffers.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6t
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/pytest timeout 600 python -m pytest tests/test_gpu_zstd.py tests/test_abi.py -m gpu -q -x -k "pinned or rolling or abi or trim"; echo "pytest rc $? $(tail -1 $OUT/pytest.log)" | tee $OUT/summary.txt
for mode in "" "--pinned"; do
  timeout 400 python tools/e2e_probe.py C2 C4 --ctx 4 --steps 12 $mode 2>/dev/null | grep config | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'], 'pinned' if d.get('pinned_buffers') else 'pageable', 'one_call', min(d['one_call_ms']), 'steady', d['two_contexts_ms_per_batch'], 'dev', d['device_resident_ms'], d['same_bytes'])" | tee -a $OUT/summary.txt
done
