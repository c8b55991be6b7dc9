#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"; HZ_WITNESS_LIB=$PWD/variants/libhz_$v.so python tools/poseidon_microbench.py 2>&1 | tail -4
done
