cd $GRAFT_REPO_ROOT
run() { echo "== $1"; ( eval "export $1"; timeout 300 python tools/experiments/masked_stream_churn.py 2 1 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200 ); }
run "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0"
run "HSA_SCRATCH_SINGLE_LIMIT=4000000000"
run "HSA_SCRATCH_SINGLE_LIMIT_ASYNC=40000000000"
run "HSA_SCRATCH_MEM_SIZE=40000000000"
run "GPU_MAX_HW_QUEUES=24"
run "HSA_NO_SCRATCH_RECLAIM=1"
