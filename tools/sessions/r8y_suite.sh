#!/bin/bash
# Session r8y: the GPU suite and the smoke on the last tree (after r8r only comments and the parts test's small shape changed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r8y
mkdir -p $OUT
cd $R
ulimit -c 0
bash tools/gpu_guard.sh $OUT/smoke timeout 400 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc $?" | tee -a $OUT/summary.txt
bash tools/gpu_guard.sh $OUT/pytest_gpu timeout 1800 python -m pytest tests -m gpu -q -x; echo "pytest rc $? $(tail -1 $OUT/pytest_gpu.log)" | tee -a $OUT/summary.txt
