#!/bin/bash
# Session r6y: what copy shape reaches the guide's 6.3 TB/s on this box (non-temporal loads / stores, grid sizes, contiguous chunks per workgroup)?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r6y
cd $R
timeout 300 tools/_build/copy_probe > gpurun_out/r6y/copy_probe.txt 2>&1
head -45 gpurun_out/r6y/copy_probe.txt
