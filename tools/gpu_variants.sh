#!/bin/bash
# bench each variant library given as argument (names under variants/), print value, ms/step and kernel times
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$PWD/variants/libhz_$v.so; [ "$v" = "base" ] && lib=$PWD/circuits_amd/libhermez_witness.so
  echo "$v: $(HZ_WITNESS_LIB=$lib python bench.py --steps ${STEPS:-4} --warmup ${WARMUP:-2} --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --distinct-batches ${DISTINCT:-4} ${BENCH_ARGS} 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)"
done
