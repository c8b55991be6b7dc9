#!/bin/bash
# value_export against the size of the export kernels' grids (HZ_EXPORT_BLOCKS): does a copy that leaves wavefront slots to the step overlap with it?
cd $GRAFT_REPO_ROOT
for b in "${@:-65536 8192 2048 1024 512}"; do
  echo "HZ_EXPORT_BLOCKS=$b: $(HZ_EXPORT_BLOCKS=$b timeout 500 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-node --no-e2e --no-withdraw --no-poseidon --no-sweep --no-deep-state 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); e=d["export"]; print("value %.0f  export_ms/batch %.2f (x4: %s)  value_export %.0f  x4 %s" % (d["value"], e["export_ms_per_batch"], e.get("export_ms_per_batch_4_per_call"), e["value_export"], e.get("value_export_4_per_call")))' 2>&1 | tail -1)"
done
