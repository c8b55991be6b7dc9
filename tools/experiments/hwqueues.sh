cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env "$2" python bench.py --steps ${3:-8} --warmup 2 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --distinct-batches 4 --no-shard $4 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["step_latency_ms"])' 2>&1 | tail -1)"; }
for i in 1 2 3; do
run default "X=1"
run ed_high "HZ_X_PRIO=-1 0 0 0 0"
run ed_high_main_low "HZ_X_PRIO=-1 0 0 1 0"
run ed_fix_fee_high "HZ_X_PRIO=-1 -1 -1 0 0"
done
