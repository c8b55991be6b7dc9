#!/bin/bash
# the driver's own command line (BENCH_rNN: --gpus 1 --steps 20 --warmup 5), wall time, value / value_e2e / value_node
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/drv
for v in "" "HZ_NODE_UV_THREADS=2"; do
( time env $v python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/drv/line.log 2>&1
grep '^{' gpurun_out/drv/line.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("'"$v"'", d["value"], d["ms_per_step"], d["value_e2e"], d.get("value_node"), d["node_host"].get("ms_per_step"), d["deep_state"]["value"], d["withdraw"]["value"])'
grep real gpurun_out/drv/line.log
done
