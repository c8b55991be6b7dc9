cd $GRAFT_REPO_ROOT
bash tools/profile.sh > gpurun_out/profile_run.log 2>&1; tail -3 gpurun_out/profile_run.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log > gpurun_out/bench_line.json; cut -c1-300 gpurun_out/bench_line.json
timeout 500 bash tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1; tail -3 gpurun_out/pmc_run.log | cut -c1-200
timeout 300 python bench.py --cpu-sample 0 --no-poseidon --batches-per-launch 1 --inflight 1 --steps 20 --warmup 5 --latency-scheduling 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("single batch, latency scheduling", d["value"], d["ms_per_step"], d["kernels_ms"])'
timeout 300 python bench.py --cpu-sample 0 --no-poseidon --batches-per-launch 1 --inflight 1 --steps 20 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("single batch", d["value"], d["ms_per_step"])'
timeout 300 python bench.py --cpu-sample 0 --workload withdraw 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("withdraw", d["value"], d["ms_per_step"], d["roofline"]["frac"])'
