#!/bin/bash
# HZ_FLAG_LATENCY contexts (CU-masked internal streams, each owning a hardware queue) with SEVERAL contexts in flight, against plain
# contexts: batches per launch x contexts in flight. (HZ_FORCE_LATENCY_SCHEDULING=1 gives every context of the process the flag.)
cd $GRAFT_REPO_ROOT
export HZ_MAX_PARTITIONED=4   # (the library partitions two contexts per device by default)
B="python bench.py --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --no-verify --distinct-batches 4"
for cfg in ${POINTS:-"1 2" "1 3" "1 4" "2 2" "2 3" "2 4" "4 2" "4 3" "4 4" "8 2" "8 3" "8 4" "16 2"}; do
  set -- $cfg
  for f in 0 1; do
    if [ $f = 1 ]; then export HZ_FORCE_LATENCY_SCHEDULING=1; else unset HZ_FORCE_LATENCY_SCHEDULING; fi
    echo "B=$1 inflight=$2 latency_flag=$f: $($B --steps ${STEPS:-12} --warmup 3 --batches-per-launch $1 --inflight $2 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
  done
done
