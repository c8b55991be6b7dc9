#!/bin/bash
# round 4: the whole GPU suite, then everything committed under profiles/ for the round
cd $GRAFT_REPO_ROOT
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=10 ) 2>&1 | tail -25
HZ_ROUND=4 bash tools/round_profile.sh 2>&1 | tail -30
