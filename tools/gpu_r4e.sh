#!/bin/bash
# round 4: the flaky 8-rank bench test (error text), tests that postdate the last full run, builder timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_multigpu.py -m gpu -x -q -k "gpus8" 2>&1 | tail -25 > gpurun_out/r4e/gpus8_$i.log; tail -1 gpurun_out/r4e/gpus8_$i.log; done
( time timeout 1200 python -m pytest tests/test_native_builder.py tests/test_derived_signals.py tests/test_reference_suites.py tests/test_batch_builder_device.py -m gpu -x -q --durations=6 ) 2>&1 | tail -20
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-sweep --no-withdraw --no-poseidon --no-deep-state --no-node > gpurun_out/r4e/bench.log 2>&1
grep '^{' gpurun_out/r4e/bench.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['config']['batch_builder']), d['config']['batch_build_s']); print(json.dumps(d.get('roofline'))[:900]); print(json.dumps(d.get('roofline_valu'))[:900])"
tail -3 gpurun_out/r4e/bench.log | cut -c1-300
