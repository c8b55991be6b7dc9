#!/bin/bash
# VALU-side counters per kernel (separate pass from any tracing other than --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/valu -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 "$@" > $OUT/bench_valu.log 2>&1
tail -2 $OUT/bench_valu.log | cut -c1-300
find $OUT -name "*.csv" | head
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc"
for f in glob.glob(out + "/valu/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[k] += 1
    with open(out + "/valu_summary.csv", "w") as o:
        names = sorted({c for k in acc for c in acc[k]})
        o.write("kernel,dispatches," + ",".join(names) + "\n")
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
            o.write(k + "," + str(n[k]) + "," + ",".join("%.4g" % (acc[k][c] / n[k]) for c in names) + "\n")
    print(open(out + "/valu_summary.csv").read())
PY
find $OUT -name "*.csv" -size +4M -delete
