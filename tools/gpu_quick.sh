#!/bin/bash
# quick GPU loop: parity tests, Poseidon microbench, one bench line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/quick; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/poseidon_microbench.py 2>&1 | tail -4
timeout 600 python bench.py --cpu-sample 0 "$@" > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])'
