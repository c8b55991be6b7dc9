#!/bin/bash
# GPU_MAX_HW_QUEUES (HIP runtime: hardware queues per process, default 4) against the latency regime: a context runs its chains on
# several streams, two contexts in flight are 8+ streams; streams that share a hardware queue run their kernels one after the other.
cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --no-verify --distinct-batches 4"
for q in default 8 16 2; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for cfg in "1 2 12" "1 4 12" "4 2 8" "32 2 4"; do
    set -- $cfg
    echo "queues=$q B=$1 inflight=$2: $($B --steps $3 --warmup 2 --batches-per-launch $1 --inflight $2 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], (d.get("single_batch_latency_ms") or {}).get("default"), (d.get("single_batch_latency_ms") or {}).get("latency_flag"))' 2>&1 | tail -1)"
  done
done
