cd $GRAFT_REPO_ROOT
export HZ_MAX_PARTITIONED=4   # (the library partitions two contexts per device by default)
export HZ_FORCE_LATENCY_SCHEDULING=1
B="python bench.py --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard --no-verify --distinct-batches 4"
for cfg in "1 1" "1 2" "1 4" "2 2" "2 4"; do
  set -- $cfg
  for v in lat nolat; do
    if [ $v = nolat ]; then export HZ_SMT_NO_LATENCY_FORM=1; else unset HZ_SMT_NO_LATENCY_FORM; fi
    echo "B=$1 inflight=$2 flagged, smt form=$v: $($B --steps 12 --warmup 3 --batches-per-launch $1 --inflight $2 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
  done
done
