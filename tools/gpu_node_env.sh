#!/bin/bash
# value_node under different process environments (which HIP runtime, runtime knobs): one bench.py run each, Node loop in sync mode
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/node
TL=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
i=0
for e in "" "LD_PRELOAD=$TL/libamdhip64.so" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=32" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
  i=$((i+1))
  HZ_NODE_ENV="$e" HZ_NODE_MODE=sync timeout 400 python bench.py --cpu-sample 0 --no-deep-state --no-withdraw --no-poseidon --distinct-batches 8 2>&1 | grep "^{" | tail -1 > gpurun_out/node/env_$i.json
  python - "$e" $i <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/node/env_%s.json" % sys.argv[2]))
    print("env [%s] value %.0f node %s" % (sys.argv[1], d["value"], d.get("value_node") or d.get("node_host")))
except Exception as ex:
    print("env [%s] failed: %s" % (sys.argv[1], ex))
PY
done
