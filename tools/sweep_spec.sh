#!/bin/bash
# Speculation-policy sweep of a match finder on the GPU box: KC_SPEC_W0 (width after a match) x KC_SPEC_GROW (0 keep, 1 +1, 2 double).
# Usage: bash tools/sweep_spec.sh <config> <gib> "w0,grow w0,grow ..."
CFG=${1:-C2}; GIB=${2:-4}; PAIRS=${3:-"1,1 1,0 2,0 2,1 1,2 4,0"}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweeps
for pr in $PAIRS; do
  w0=${pr%,*}; gr=${pr#*,}
  KC_SPEC_W0=$w0 KC_SPEC_GROW=$gr python bench.py --config $CFG --gib $GIB --steps 3 --warmup 1 --no-cpu-baseline --no-device-verify --no-end-to-end 2>/dev/null | tail -1 | \
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$CFG w0=$w0 grow=$gr  kernel_ms=%.2f entropy_ms=%.2f ms_per_step=%.2f MB/s=%.0f ratio=%.5f' % (j['roofline']['kernel_ms'], j['roofline']['entropy_kernel_ms'], j['ms_per_step'], j['value'], j['ratio']))" | tee -a gpurun_out/sweeps/$CFG.txt
done
