#!/bin/bash
# flagged contexts in flight (one batch each) with k_smt's latency form forced on (HZ_SMT_LATENCY_FORM=1), at one (tree) and two
# (variants/libhz_smtw2.so) wavefronts per SIMD, against the default (plain form). usage: bash tools/experiments/flagged_latform.sh
mkdir -p gpurun_out; out=gpurun_out/flagged_latform.txt; : > $out
run() {  # name lib env inflight
  local line=$(HZ_WITNESS_LIB=$2 env $3 timeout 300 python bench.py --steps 24 --warmup 4 --cpu-sample 0 --no-poseidon --no-export --no-withdraw --no-e2e --no-deep-state --no-sweep --no-node --batches-per-launch 1 --inflight $4 --latency-scheduling --distinct-batches 8 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$1, $4 contexts: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)" >> $out
}
base=$PWD/circuits_amd/libhermez_witness.so; w2=$PWD/variants/libhz_smtw2.so
for r in 1 2; do
  for i in 1 2 4; do
    run "plain form" $base "HZ_X=0" $i
    run "latency form, 1 wavefront per SIMD" $base "HZ_SMT_LATENCY_FORM=1" $i
    run "latency form, 2 wavefronts per SIMD" $w2 "HZ_SMT_LATENCY_FORM=1" $i
  done
done
cat $out
