#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then HBM counters in
# separate passes. Summaries are copied from gpurun_out/ into profiles/ by hand (see profiles/README.md).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python bench.py --steps 4 --warmup 1 --inflight 1 --cpu-sample 0 > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- python bench.py --steps 2 --warmup 1 --inflight 1 --cpu-sample 0 > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- python bench.py --steps 2 --warmup 1 --inflight 1 --cpu-sample 0 > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -30
# keep the merge small: drop raw per-dispatch traces above 8 MiB
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT/*/* | head -40
