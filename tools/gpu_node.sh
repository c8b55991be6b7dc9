#!/bin/bash
# value_node with 1 and 2 libuv pool threads (the addon's step() runs on the pool): bench lines under gpurun_out/node/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/node
for t in "$@"; do
  HZ_NODE_MODE=${t#*:} HZ_NODE_UV_THREADS=${t%%:*} timeout 400 python bench.py --cpu-sample 0 --no-deep-state --no-withdraw --no-poseidon --distinct-batches 8 2>&1 | grep "^{" | tail -1 > gpurun_out/node/line_${t/:/_}.json
  python - ${t/:/_} <<'PY'
import json, sys
d = json.load(open("gpurun_out/node/line_%s.json" % sys.argv[1]))
print("uv threads", sys.argv[1], "value", d["value"], "e2e", d.get("value_e2e"), "node", d.get("value_node"), d.get("node_host"))
PY
done
