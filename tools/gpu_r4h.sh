#!/bin/bash
# round 4, final: the whole GPU suite, then the committed profile set without the microbenchmarks
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) 2>&1 | tail -16
HZ_ROUND=4 SKIP_MICRO=1 timeout 1500 bash tools/round_profile.sh 2>&1 | tail -12
