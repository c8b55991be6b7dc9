#!/bin/bash
# HZ_LDS_PAD (unused dynamic LDS per wavefront: at most 160 KB / pad wavefronts per compute unit, whatever kernels they belong to) against
# the CU partition of HZ_FLAG_LATENCY contexts, in the latency regime
cd $GRAFT_REPO_ROOT   # (needs the hz_lds_pad() launch-site patch of this experiment: see profiles/r05_latency_regime.txt)
B="python bench.py --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --no-verify --distinct-batches 4"
for cfg in ${POINTS:-"1 1" "1 2" "1 4" "2 2" "2 4" "4 2" "4 4" "8 2"}; do
  set -- $cfg
  for v in 0 40960 20480 flag; do
    extra=""
    if [ $v = flag ]; then unset HZ_LDS_PAD; extra="--latency-scheduling"; elif [ $v = 0 ]; then unset HZ_LDS_PAD; else export HZ_LDS_PAD=$v; fi
    echo "B=$1 inflight=$2 pad=$v: $($B $extra --steps 12 --warmup 3 --batches-per-launch $1 --inflight $2 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
  done
done
