#!/bin/bash
# like gpu_variants.sh, for variants that compute wrong values on purpose (knock-outs): constraint failures are ignored
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$PWD/variants/libhz_$v.so; [ "$v" = "base" ] && lib=$PWD/circuits_amd/libhermez_witness.so
  echo "$v: $(HZ_WITNESS_LIB=$lib python tools/experiments/bench_nocheck.py --steps ${STEPS:-4} --warmup ${WARMUP:-2} --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --distinct-batches ${DISTINCT:-4} --no-verify --no-shard ${BENCH_ARGS} 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)"
done
