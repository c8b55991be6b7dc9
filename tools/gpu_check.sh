#!/bin/bash
# Round check on the GPU box: parity tests, default bench under rocprofv3 --stats, inflight/B sweep.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/check
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log | cut -c1-600
for cfg in "16 2" "16 3" "8 4" "32 2"; do
  set -- $cfg
  timeout 600 python bench.py --steps 6 --warmup 1 --batches-per-launch $1 --inflight $2 --cpu-sample 0 > $OUT/bench_B$1_I$2.log 2>&1
  echo "B=$1 inflight=$2: $(tail -1 $OUT/bench_B$1_I$2.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)"
done
