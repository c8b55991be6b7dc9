#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/fixg; mkdir -p $OUT; rm -f $OUT/fixg.log
for g in 8 6; do
  echo "== HZ_ED_FIX_G=$g" | tee -a $OUT/fixg.log
  HZ_ED_FIX_G=$g timeout 1500 python -m pytest tests/test_witness_gpu.py -m gpu -x -q -k "throughput" 2>&1 | tail -3 | tee -a $OUT/fixg.log
done
for rep in 1 2; do
for g in 4 8 6; do
  echo "fix_g=$g: $(HZ_ED_FIX_G=$g python bench.py --steps 8 --warmup 3 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-node --no-deep-state --distinct-batches 8 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)" | tee -a $OUT/fixg.log
done
done
