#!/bin/bash
# round 4, first GPU call: the new parity tests (adversarial fuzz, headline launch whole buffer, per-instance failures) + mulbench
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4a; rm -rf $OUT; mkdir -p $OUT
(cd tools/microbench && timeout 300 ./mulbench | python check_v8.py) > $OUT/mulbench.txt 2>&1; tail -30 $OUT/mulbench.txt
timeout 1500 python -m pytest tests/test_adversarial_fuzz.py tests/test_witness_gpu.py -m gpu -x -q --durations=12 -k "fuzz or headline or failure or constraint" 2>&1 | tail -40
