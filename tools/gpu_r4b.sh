#!/bin/bash
# round 4: the whole GPU suite, then the default bench line (with the new occupancy points)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4b; rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) 2>&1 | tail -32
( time timeout 900 python bench.py ) > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench_line.json
tail -5 $OUT/bench.log | cut -c1-600
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4b/bench_line.json"))
for k in ("value", "ms_per_step", "value_e2e", "value_node", "value_deep_state", "single_batch_latency_ms", "batches_sweep", "roofline_valu"):
    print(k, d.get(k))
print("roofline", {k: v for k, v in d["roofline"].items() if k != "note"})
print("kernels_ms", d["kernels_ms"])
PY
