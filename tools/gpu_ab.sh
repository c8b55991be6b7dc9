#!/bin/bash
# A/B on ONE box: the GPU parity suite on the library in the tree, then variants/libhz_$1.so against it (bench without the side lines)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.log
for rep in 1 2; do
  STEPS=${STEPS:-8} WARMUP=3 DISTINCT=${DISTINCT:-8} bash tools/gpu_variants.sh "$@" base 2>&1 | tee -a $OUT/ab.log
done
