#!/bin/bash
# plain (unpartitioned) contexts whose internal streams each own a hardware queue (HZ_DEDICATED_QUEUES=1: CU mask of every CU) against
# the default (streams share the process's hardware queues)
cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --no-verify --distinct-batches 4"
for cfg in "32 2 6" "16 2 8" "8 2 10" "1 2 12"; do
  set -- $cfg
  for v in 0 1 1e 0 1; do
    if [ $v = 0 ]; then unset HZ_DEDICATED_QUEUES; else export HZ_DEDICATED_QUEUES=$v; fi
    echo "B=$1 inflight=$2 dedicated=$v: $($B --steps $3 --warmup 2 --batches-per-launch $1 --inflight $2 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
  done
done
