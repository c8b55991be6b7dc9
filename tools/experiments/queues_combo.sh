#!/bin/bash
# headline step: default queues against GPU_MAX_HW_QUEUES=16 and / or a hardware queue of its own for the ladder stream (HZ_DEDICATED_QUEUES=1e)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --no-verify --distinct-batches 4"
for i in 1 2 3; do
for v in base q16 ded both; do
  unset GPU_MAX_HW_QUEUES HZ_DEDICATED_QUEUES
  case $v in q16) export GPU_MAX_HW_QUEUES=16;; ded) export HZ_DEDICATED_QUEUES=1e;; both) export GPU_MAX_HW_QUEUES=16 HZ_DEDICATED_QUEUES=1e;; esac
  echo "$v: $($B 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
done; done
