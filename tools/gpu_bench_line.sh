#!/bin/bash
# the default bench on the current tree, the line's headline fields
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python bench.py "$@" ) > gpurun_out/final_bench.log 2>&1
grep "^{" gpurun_out/final_bench.log | tail -1 | python -c '
import json, sys
d = json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d.get("value_deep_state"), d.get("value_e2e"), d.get("value_node"), d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"].get("frac_valu"),
      d.get("cpu_baseline", {}).get("value"), d["config"].get("batch_builder", {}).get("ms_per_batch"))'
grep real gpurun_out/final_bench.log
