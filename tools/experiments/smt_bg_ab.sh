#!/bin/bash
# A/B of the external constant-block writer (k_smt_bg, HZ_SMT_BG_ON) against BgZero inside k_smt: profiles/r05_ksmt_bg_writer.txt
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python bench.py --no-node --cpu-sample 0 --no-withdraw --no-poseidon --no-sweep --no-deep-state --no-export --no-e2e --steps 10 --warmup 3 --no-verify 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"].get("smt"), d["kernels_ms"].get("smt_bg"))')"; }
for r in 1 2; do
run "A=1"
run "HZ_SMT_BG_ON=1"
run "HZ_SMT_BG_ON=1 HZ_SMT_BG_GRID=128"
run "HZ_SMT_BG_ON=1 HZ_SMT_BG_GRID=1024"
done
