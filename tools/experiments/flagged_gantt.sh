#!/bin/bash
# kernel Gantt of flagged contexts in flight (HZ_FLAG_LATENCY), one batch each: where does a 4.9 ms step of two 8.5 ms chains go?
# usage (GPU box): bash tools/experiments/flagged_gantt.sh [contexts] -> gpurun_out/flagged_gantt.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fl_tl; rm -rf $OUT; mkdir -p $OUT; cd $R
I=${1:-2}
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python bench.py --steps 12 --warmup 3 --cpu-sample 0 --no-verify --no-poseidon --no-export --no-withdraw --no-e2e --no-deep-state --no-sweep --no-node --batches-per-launch 1 --inflight $I --latency-scheduling --distinct-batches 4 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/flagged_gantt.txt <<'PY'
import csv, subprocess, sys
rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
t0 = rows[0][0]
fr = [(s - t0) / 1e6 for s, n in rows if "k_main_front" in n]
lo = fr[len(fr) // 2] - 0.5
print("# kernel Gantt of flagged contexts in flight, one batch each (0.2 ms per column; digit = launches of that kernel running); k_main_front launches at ms:", [round(x - lo, 2) for x in fr[len(fr) // 2: len(fr) // 2 + 8]])
sys.stdout.flush()
subprocess.run([sys.executable, "tools/gantt.py", sys.argv[1], str(lo), str(lo + 24), "0.2"])
PY
rm -rf $OUT
cat $R/gpurun_out/flagged_gantt.txt | cut -c1-160
