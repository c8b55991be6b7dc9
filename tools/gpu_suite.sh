#!/bin/bash
# the whole GPU suite and the smoke test in one gpurun call
cd $GRAFT_REPO_ROOT
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=5 ) 2>&1 | tail -14
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
