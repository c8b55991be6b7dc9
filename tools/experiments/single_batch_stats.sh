#!/bin/bash
# rocprofv3 kernel statistics of ONE 2048-transaction batch alone on the device (HZ_FLAG_LATENCY): the two critical chains kernel by kernel.
# usage (GPU box): bash tools/experiments/single_batch_stats.sh -> gpurun_out/single_batch_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sb; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python bench.py --steps 16 --warmup 3 --cpu-sample 0 --no-verify --no-poseidon --no-export --no-withdraw --no-e2e --no-deep-state --no-sweep --no-node --batches-per-launch 1 --inflight 1 --latency-scheduling --distinct-batches 4 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-160
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --batches-per-launch 1 --inflight 1 --latency-scheduling --steps 16 ... (round 6, final kernels): ONE 2048-transaction batch alone on the device, HZ_FLAG_LATENCY"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("name,calls,average_us,total_us,percentage")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"].split("(")[0].replace("void hz::", "").replace("hz::", "")
    print("%s,%s,%.1f,%.0f,%s" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
} > $R/gpurun_out/single_batch_kernel_stats.csv
rm -rf $OUT
head -30 $R/gpurun_out/single_batch_kernel_stats.csv
