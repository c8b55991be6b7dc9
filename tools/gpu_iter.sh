#!/bin/bash
# one iteration on the GPU box: a subset of the parity tests (-k "$1"), then the default bench without the CPU sample
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/iter; rm -rf $OUT; mkdir -p $OUT
K="$1"; shift
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -4
timeout 600 python bench.py --cpu-sample 0 "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("value_e2e"), d["kernels_ms"]); print("withdraw", d.get("withdraw")); print("roofline", d["roofline"])'
