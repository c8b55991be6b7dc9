#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): rocprofv3 kernel trace + stats of the default bench command,
# then the HBM counters in two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass). The summaries
# written to gpurun_out/prof/ are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $R
ARGS="--steps 4 --warmup 1 --cpu-sample 0 --no-e2e --no-deep-state --no-sweep --no-export --distinct-batches 4"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python bench.py $ARGS --calibrate-copy > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python bench.py $ARGS --calibrate-copy > $OUT/bench_write.log 2>&1
python tools/profile_summary.py $OUT "python bench.py $ARGS"
tail -1 $OUT/bench_trace.log | cut -c1-400
# keep the merge small: drop raw per-dispatch traces
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
