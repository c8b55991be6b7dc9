#!/bin/bash
# Everything this repository runs on the GPU box through gpurun, one entry point:  gpurun -- 'bash tools/gpu_suite.sh <what> [args]'
#   suite                  the whole GPU parity suite and the smoke test (default)
#   iter "<-k expr>" ...   a subset of the parity tests, then the bench without the CPU sample (bench arguments follow)
#   bench ...              the default bench line and its headline fields
#   variants a b ..        bench each library under variants/libhz_<name>.so ("base" = the tree's), STEPS / WARMUP / DISTINCT / BENCH_ARGS,
#                          NOCHECK=1 for knock-outs that compute wrong values on purpose
#   sweep "B I" ..         batches-per-launch x contexts-in-flight points
#   timeline ...           kernel trace of a short bench run -> Gantt of the timed region (gpurun_out/tl)
#   pmc ...                SQ_INSTS_VALU & co per kernel, a --pmc pass of its own (gpurun_out/pmc)
#   pmc-poseidon           the same for the Poseidon batch kernels (gpurun_out/pmc_pos)
#   micro                  the microbenchmarks of tools/microbench (built by tools/microbench/build.sh before the call)
# (the round's committed profile set: tools/round_profile.sh; HBM counters: tools/profile.sh, tools/pmc_only.sh, tools/pmc_diag.sh)
cd $GRAFT_REPO_ROOT
what=${1:-suite}; shift
fields='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("value_deep_state"), d.get("value_e2e"), d.get("value_node"), d.get("export_ms_per_batch"), d.get("value_export"), d["roofline"]["frac"], d["roofline"].get("frac_valu"), d["kernels_ms"])'
case $what in
suite)
  ( time timeout 2300 python -m pytest tests -m gpu -x -q --durations=10 ) 2>&1 | tail -14
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
iter)
  K="$1"; shift
  timeout 1200 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -4
  mkdir -p gpurun_out/iter
  timeout 600 python bench.py --cpu-sample 0 "$@" > gpurun_out/iter/bench.log 2>&1
  grep '^{' gpurun_out/iter/bench.log | tail -1 | python -c "$fields" ;;
bench)
  mkdir -p gpurun_out
  ( time python bench.py "$@" ) > gpurun_out/final_bench.log 2>&1
  grep '^{' gpurun_out/final_bench.log | tail -1 > gpurun_out/bench_line.json
  python -c "$fields" < gpurun_out/bench_line.json; grep real gpurun_out/final_bench.log ;;
variants)
  script=bench.py; extra=""
  [ -n "$NOCHECK" ] && { script=tools/experiments/bench_nocheck.py; extra="--no-verify --no-shard"; }
  for v in "$@"; do
    lib=$PWD/variants/libhz_$v.so; [ "$v" = "base" ] && lib=$PWD/circuits_amd/libhermez_witness.so
    echo "$v: $(HZ_WITNESS_LIB=$lib python $script --steps ${STEPS:-4} --warmup ${WARMUP:-2} --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-export --distinct-batches ${DISTINCT:-4} $extra ${BENCH_ARGS} 2>&1 | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)"
  done ;;
sweep)
  mkdir -p gpurun_out/sweep
  for cfg in "$@"; do
    set -- $cfg
    timeout 600 python bench.py --steps ${3:-6} --warmup 2 --batches-per-launch $1 --inflight $2 --cpu-sample 0 --no-verify --no-export > gpurun_out/sweep/bench_B$1_I$2.log 2>&1
    echo "B=$1 inflight=$2: $(grep '^{' gpurun_out/sweep/bench_B$1_I$2.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
  done ;;
timeline)
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TL_OUT:-tl}; rm -rf $OUT; mkdir -p $OUT; cd $R
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-verify --no-poseidon --no-export "$@" > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-300
  f=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $f > $OUT/timeline.txt
  gzip -9 -c $f > $OUT/kernel_trace.csv.gz
  find $OUT -name "*.csv" -size +1M -delete; ls -la $OUT ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT; cd $R
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/valu -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-sweep --no-export "$@" > $OUT/bench_valu.log 2>&1
  tail -2 $OUT/bench_valu.log | cut -c1-300
  python tools/pmc_summary.py valu "$@"
  find $OUT -name "*.csv" -size +4M -delete ;;
pmc-poseidon)
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_pos; rm -rf $OUT; mkdir -p $OUT; cd $R
  CMD="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-sweep --no-e2e --no-deep-state --no-withdraw --no-node --no-export --distinct-batches 1 --batches-per-launch 1 --inflight 1"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/run -o run --output-format csv -- $CMD > $OUT/bench.log 2>&1
  tail -1 $OUT/bench.log | cut -c1-200
  python tools/pmc_summary.py poseidon "$CMD"
  find $OUT -name "*.csv" -size +2M -delete ;;
micro)
  OUT=gpurun_out/micro; mkdir -p $OUT
  for b in instbench mulbench invbench storebench; do [ -x tools/microbench/$b ] || echo "tools/microbench/$b not built (tools/microbench/build.sh)"; done
  for b in ${@:-mulbench}; do timeout 300 tools/microbench/$b > $OUT/$b.txt 2>&1; tail -40 $OUT/$b.txt; done ;;
*) echo "unknown: $what"; exit 2 ;;
esac
