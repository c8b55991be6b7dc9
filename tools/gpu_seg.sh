#!/bin/bash
# experiment: segment-lane lockstep ladder (HZ_ED_SEG_G) -- parity of the throughput tests under each setting, then step times on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/seg; mkdir -p $OUT
for g in 0 4 3 2; do
  echo "== HZ_ED_SEG_G=$g" | tee -a $OUT/seg.log
  HZ_ED_SEG_G=$g timeout 1500 python -m pytest tests/test_witness_gpu.py -m gpu -x -q -k "throughput" 2>&1 | tail -3 | tee -a $OUT/seg.log
done
for rep in 1 2; do
for g in 0 4 3 2; do
  echo "seg_g=$g: $(HZ_ED_SEG_G=$g python bench.py --steps 8 --warmup 3 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-node --no-deep-state --distinct-batches 8 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)" | tee -a $OUT/seg.log
done
done
