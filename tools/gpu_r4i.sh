#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_smt_kat.py tests/test_witness_gpu.py -m gpu -x -q -k "not headline and not config4 and not config5 and not throughput" ) 2>&1 | tail -8
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-sweep --no-withdraw --no-poseidon --no-node --no-e2e > gpurun_out/r4i_bench.log 2>&1
grep '^{' gpurun_out/r4i_bench.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value',d['value'], d['ms_per_step'], 'deep', d.get('value_deep_state')); print(d['kernels_ms']); print(d['roofline']['launch_ms'], d['roofline']['frac'], d.get('deep_state',{}).get('kernels_ms',{}).get('smt'))"
tail -3 gpurun_out/r4i_bench.log | cut -c1-300
