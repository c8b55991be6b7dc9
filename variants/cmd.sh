cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --cpu-sample 0 --no-poseidon 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])'
timeout 300 python bench.py --cpu-sample 0 --no-poseidon --batches-per-launch 1 --inflight 1 --steps 20 --warmup 5 --latency-scheduling 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("single batch", d["value"], d["ms_per_step"], d["kernels_ms"])'
