#!/bin/bash
# Session r6l: ring geometry of the SpeedBetter match finder (2 KiB ring / 768 ahead; 1 KiB / 640 ahead) against the default on C5.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SESSION=r6l CONFIG=C5 TAGS="rb2k base ah512 rb2k base ah512" NO_PYTEST=1
cd $R
bash tools/sessions/r5e_ab.sh
