#!/bin/bash
# Session r6k: the SpeedBetter match finder with the LDS source ring: parity subset (level 3, dictionaries, streams), then C5 same-box
# A/B against the build without the ring (KC_LIB_TAG=noring: -DZB_RING=0).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SESSION=r6k CONFIG=C5 TAGS="noring base noring base"
export PYTEST_K="corpus_units or edge or stress or ragged or long_units or randomized_options or dictionary or streams_with_flush or rolling"
cd $R
bash tools/sessions/r5e_ab.sh
