#!/bin/bash
# Session r6m: the S2 encoder with the LDS source ring (139 VGPRs: 12 waves per CU), the same held to 128 VGPRs (S2_WPE=4), and without
# the ring, on C4; parity subset of the S2 suite first.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
mkdir -p gpurun_out/r6m
timeout 900 python -m pytest tests/test_gpu_s2.py -x -q -m gpu -k "not best" > gpurun_out/r6m/pytest_s2.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6m/pytest_s2.log)"
export SESSION=r6m CONFIG=C4 TAGS="s2noring base s2wpe4 s2noring base s2wpe4" NO_PYTEST=1
bash tools/sessions/r5e_ab.sh
