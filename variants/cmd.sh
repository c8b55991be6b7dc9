cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_poseidon.py tests/test_witness_gpu.py -m gpu -x -q -k "not config4" 2>&1 | tail -2
STEPS=4 WARMUP=2 timeout 900 bash tools/gpu_variants.sh base rowwise base rowwise
echo latency; BENCH_ARGS="--batches-per-launch 1 --inflight 1" STEPS=20 WARMUP=5 timeout 600 bash tools/gpu_variants.sh base rowwise
