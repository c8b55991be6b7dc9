#!/bin/bash
# Attribution pass: which counters does rocprofv3 expose on this box, then instruction-cache / scalar-cache /
# wait counters per kernel in separate bounded --pmc passes (kernel trace only beside them). Output: gpurun_out/diag/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/diag; rm -rf $OUT; mkdir -p $OUT
cd $R
T=${PMC_TIMEOUT:-240}
( rocprofv3 -L || rocprofv3 --list-avail ) > $OUT/counters_list.txt 2>&1
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-withdraw --distinct-batches 4 ${DIAG_BENCH_ARGS}"
python - "$OUT" > $OUT/groups.txt <<'PY'
import re, sys
txt = open(sys.argv[1] + "/counters_list.txt").read()
avail = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", txt))
want = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"],
    ["SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"],
    ["SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQC_DCACHE_MISSES_DUPLICATE", "SQC_TC_REQ", "SQC_TC_INST_REQ", "SQC_TC_DATA_READ_REQ", "SQC_TC_STALL"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM", "SQ_INST_CYCLES_VMEM_WR"],
    ["SQ_WAIT_INST_LDS", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_WAVES_EQ_64", "SQ_THREAD_CYCLES_VALU", "SQ_VALU_MFMA_BUSY_CYCLES"],
    ["TCC_EA_WRREQ_STALL_sum", "TCC_EA_WRREQ_sum", "TCC_EA_WRREQ_64B_sum", "TCC_REQ_sum", "TCC_WRITE_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TA_DATA_STALL_CYCLES_sum"],
]
for g in want:
    g = [c for c in g if c in avail]
    if g:
        print(" ".join(g))
PY
cat $OUT/groups.txt
i=0
while read -r grp; do
  i=$((i+1))
  timeout $T rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o run -- python bench.py $ARGS > $OUT/bench_p$i.log 2>&1; echo "pass $i rc=$? ($grp)"
done < $OUT/groups.txt
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set)); grid = {}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("hz::", "").replace("void ", "")
        g = int(r.get("Grid_Size", 0) or 0)
        grid[k] = max(grid.get(k, 0), g)
rows = []
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("hz::", "").replace("void ", "")
        if int(r.get("Grid_Size", 0) or 0) != grid[k]:
            continue   # the largest-grid dispatches only (the transaction launch)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
with open(out + "/diag_summary.csv", "w") as o:
    o.write("# per-dispatch means over the largest-grid dispatches of each kernel; separate --pmc passes of: python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-withdraw --distinct-batches 4\n")
    o.write("kernel,grid," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0) / max(1, len(n[k].get("SQ_WAVE_CYCLES", [1])))):
        o.write(k + "," + str(grid[k]) + "," + ",".join(("%.4g" % (acc[k][c] / len(n[k][c]))) if c in acc[k] else "" for c in names) + "\n")
print(open(out + "/diag_summary.csv").read()[:6000])
PY
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*kernel_trace.csv" -delete
