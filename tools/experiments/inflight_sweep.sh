#!/bin/bash
# contexts in flight x batches per launch at a constant ~64 resident batches: does a third / fourth context hide more of the serial tails?
# usage (GPU box): bash tools/experiments/inflight_sweep.sh [rounds]  -> gpurun_out/inflight_sweep.txt
set -u
mkdir -p gpurun_out
out=gpurun_out/inflight_sweep.txt
: > $out
common="--steps 12 --warmup 3 --no-sweep --no-deep-state --no-node --no-export --no-withdraw --no-poseidon --cpu-sample 0 --no-e2e"
for r in $(seq 1 ${1:-2}); do
  for cfg in "2 32" "3 21" "4 16"; do
    set -- $cfg
    line=$(timeout 600 python bench.py --inflight $1 --batches-per-launch $2 $common 2>/dev/null | grep '^{"metric"' | tail -1)
    python - "$1" "$2" <<PY >> $out
import json, sys
d = json.loads('''$line''' or '{}')
print("contexts %s x %s batches: %s tx/s, %s ms per step, builder phases %s" % (sys.argv[1], sys.argv[2], d.get("value"), d.get("ms_per_step"),
      (d.get("config", {}).get("batch_builder") or {}).get("phases_ms_per_batch")))
PY
  done
done
cat $out
