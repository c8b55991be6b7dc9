#!/bin/bash
# one batch alone (HZ_FLAG_LATENCY | HZ_FLAG_SOLO): the split signature prologue against the one-kernel prologue (HZ_ED_NO_PRE_SPLIT=1).
# (A build that piped the SHA-256 block expansion over the fee stream for partitioned contexts too measured 9.10-9.15 ms against 7.86-7.89:
# not in the tree.)
cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --distinct-batches 2 --batches-per-launch 1 --inflight 1 --latency-scheduling --solo --steps 40 --warmup 5"
for i in 1 2; do
for v in base nosplit; do
  unset HZ_ED_NO_PRE_SPLIT
  case $v in nosplit) export HZ_ED_NO_PRE_SPLIT=1;; esac
  echo "$v: $($B 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
done; done
