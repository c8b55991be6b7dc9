#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_declared_signals.py -m gpu -x -q --durations=5 ) 2>&1 | tail -60
