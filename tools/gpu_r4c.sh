#!/bin/bash
# round 4: derived signals, the multi-GPU readiness tests, the node facade
cd $GRAFT_REPO_ROOT
( time timeout 1700 python -m pytest tests/test_derived_signals.py tests/test_node.py tests/test_multigpu.py tests/test_poseidon.py -m gpu -x -q --durations=12 ) 2>&1 | tail -40
