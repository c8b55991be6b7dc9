#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl2; rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-verify "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f > $OUT/timeline.txt
wc -l $OUT/timeline.txt
find $OUT -name "*.csv" -size +6M -delete
