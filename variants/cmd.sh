cd $GRAFT_REPO_ROOT
STEPS=6 WARMUP=2 timeout 900 bash tools/gpu_variants.sh base nt base nt
for v in base nt base nt; do lib=$PWD/variants/libhz_$v.so; [ "$v" = "base" ] && lib=$PWD/circuits_amd/libhermez_witness.so
echo "withdraw $v: $(HZ_WITNESS_LIB=$lib timeout 300 python bench.py --cpu-sample 0 --workload withdraw 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["frac"])')"; done
HZ_WITNESS_LIB=$PWD/variants/libhz_nt.so timeout 600 python -m pytest tests/test_witness_gpu.py -m gpu -x -q 2>&1 | tail -2
