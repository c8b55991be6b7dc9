cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
STEPS=6 WARMUP=2 timeout 900 bash tools/gpu_variants.sh base mont base mont
timeout 200 python tools/poseidon_microbench.py 2>&1 | tail -4
