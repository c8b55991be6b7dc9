#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $2 python bench.py --steps ${3:-8} --warmup 3 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-node --no-deep-state --distinct-batches 8 $4 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["batches_per_launch"], d["config"]["contexts_in_flight"])' 2>&1 | tail -1)"; }
for r in 1 2; do
run "32x2" "A=1" 8 "--batches-per-launch 32 --inflight 2"
run "64x1" "A=1" 4 "--batches-per-launch 64 --inflight 1"
run "21x3" "A=1" 12 "--batches-per-launch 21 --inflight 3"
run "16x4" "A=1" 16 "--batches-per-launch 16 --inflight 4"
run "32x2 q8" "GPU_MAX_HW_QUEUES=8" 8 "--batches-per-launch 32 --inflight 2"
run "32x2 q32" "GPU_MAX_HW_QUEUES=32" 8 "--batches-per-launch 32 --inflight 2"
done
