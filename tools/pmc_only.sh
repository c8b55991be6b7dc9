#!/bin/bash
# the two HBM counter passes and the VALU pass alone (the kernel trace of tools/profile.sh must already be in gpurun_out/prof/trace
# or is taken again when missing); every pass bounded: a counter pass that hangs is killed after PMC_TIMEOUT seconds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
T=${PMC_TIMEOUT:-200}
mkdir -p $OUT
cd $R
ARGS="--steps 4 --warmup 1 --cpu-sample 0 --no-e2e --no-deep-state --no-sweep --no-export --distinct-batches 4"
[ -d $OUT/trace ] || timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python bench.py $ARGS --calibrate-copy > $OUT/bench_fetch.log 2>&1; echo "fetch rc=$?"
timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python bench.py $ARGS --calibrate-copy > $OUT/bench_write.log 2>&1; echo "write rc=$?"
python tools/profile_summary.py $OUT "python bench.py $ARGS" | tail -3
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
