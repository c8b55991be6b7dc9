#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "$@"; do
  echo "$v: $(HZ_WITNESS_LIB=$PWD/variants/libhz_$v.so python bench.py --workload withdraw --steps 2 --warmup 1 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"], d["whole_launch"]["launch_ms"])')"
done; done
