#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_declared_signals.py tests/test_derived_signals.py tests/test_witness_gpu.py tests/test_multigpu.py tests/test_gadget_mains.py tests/test_reference_kats.py -m gpu -x -q --durations=8 ) 2>&1 | tail -40
