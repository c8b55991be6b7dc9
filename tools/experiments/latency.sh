cd $GRAFT_REPO_ROOT
run() { echo "$1: $(python bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --distinct-batches 1 --no-shard --inflight 1 $2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["step_latency_ms"])' 2>&1 | tail -1)"; }
for i in 1 2; do
run b1 "--batches-per-launch 1"
run b1_lat "--batches-per-launch 1 --latency-scheduling"
run b4 "--batches-per-launch 4"
run b4_lat "--batches-per-launch 4 --latency-scheduling"
run b32 "--batches-per-launch 32"
done
