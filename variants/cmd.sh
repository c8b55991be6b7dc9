cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/b1.log 2>&1; tail -1 gpurun_out/b1.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"], d["roofline"])'
timeout 600 python bench.py --cpu-sample 0 --workload withdraw > gpurun_out/b2.log 2>&1; tail -1 gpurun_out/b2.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("kernels_ms"), d["roofline"])'
timeout 300 python bench.py --cpu-sample 0 --batches-per-launch 1 --inflight 1 --steps 20 --warmup 5 > gpurun_out/b3.log 2>&1; tail -1 gpurun_out/b3.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])'
