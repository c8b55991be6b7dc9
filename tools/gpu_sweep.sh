#!/bin/bash
# sweep batches-per-launch x contexts-in-flight: "B I" pairs as arguments
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sweep; mkdir -p $OUT
for cfg in "$@"; do
  set -- $cfg
  timeout 600 python bench.py --steps ${3:-6} --warmup 2 --batches-per-launch $1 --inflight $2 --cpu-sample 0 --no-verify > $OUT/bench_B$1_I$2.log 2>&1
  echo "B=$1 inflight=$2: $(tail -1 $OUT/bench_B$1_I$2.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
done
