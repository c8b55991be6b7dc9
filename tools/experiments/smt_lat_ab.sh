cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "fuzz or latency or smt or rollup_main or rollup_tx" 2>&1 | tail -4
B="python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --no-node --no-deep-state --no-sweep --no-shard"
for i in 1 2; do
for v in lat nolat; do
  if [ $v = nolat ]; then export HZ_SMT_NO_LATENCY_FORM=1; else unset HZ_SMT_NO_LATENCY_FORM; fi
  echo "$v: $($B 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"]["fee_smt"], d.get("single_batch_latency_ms",{}).get("latency_flag"))')"
done; done
