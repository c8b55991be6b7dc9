#!/bin/bash
# kernel trace of a short default bench run -> per-kernel Gantt of the timed region (gpurun_out/tl/gantt.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TL_OUT:-tl}; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-verify --no-poseidon "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f > $OUT/timeline.txt
python - "$f" > $OUT/span.txt <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
t0 = rows[0][0]
print("first", 0, "last", (rows[-1][1] - t0) / 1e6)
PY
cat $OUT/span.txt
gzip -9 -c $f > $OUT/kernel_trace.csv.gz
find $OUT -name "*.csv" -size +1M -delete
ls -la $OUT
