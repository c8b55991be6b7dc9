#!/bin/bash
# A/B of two libraries on one box, alternating: the step (32 batches x 2 contexts) and one batch alone (plain, HZ_FLAG_LATENCY, | HZ_FLAG_SOLO)
# usage (GPU box): bash tools/experiments/ab_latency.sh "<variant> .." [rounds]   (variants/libhz_<variant>.so against the tree's library)
vs=$1; rounds=${2:-2}
mkdir -p gpurun_out
out=gpurun_out/ab_latency_$(echo $vs | tr ' ' '_').txt; : > $out
for r in $(seq 1 $rounds); do
  for name in $vs base; do
    lib=$PWD/variants/libhz_$name.so; [ "$name" = "base" ] && lib=$PWD/circuits_amd/libhermez_witness.so
    HZ_WITNESS_LIB=$lib timeout 900 python bench.py --steps 12 --warmup 3 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-deep-state 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['single_batch_latency_ms']
sw = {e['batches_per_launch']: e for e in d['batches_sweep']}
print('$name: step %.3f ms (%.0f tx/s); one batch %.2f / flag %.2f / solo %.2f ms; eddsa %.2f fix %.2f (flag: %.2f / %.2f); 1x2 flagged %s, 1x4 %s, 2x4 %s' % (
    d['ms_per_step'], d['value'], s['default'], s['latency_flag'], s['latency_solo_flags'], d['kernels_ms']['eddsa'], d['kernels_ms']['eddsa_fix'],
    s['kernels_ms_latency_flag']['eddsa'], s['kernels_ms_latency_flag']['eddsa_fix'],
    (sw[1].get('latency_flag_x2') or {}).get('tx_per_s'), (sw[1].get('latency_flag_x4') or {}).get('tx_per_s'), (sw[2].get('latency_flag_x4') or {}).get('tx_per_s')))
" >> $out
  done
done
cat $out
