#!/bin/bash
# Same-box A/B of whole trees: the round-4 tree against the round-5 head, alternating, then the launch-structure
# commits between them, then this tree with and without the constant marks / the rotation. The trees are `git archive`s built under
# variants/ab/<name>/ (tools/experiments/regression_ab_build.sh); each runs ITS OWN bench.py and library.
#   gpurun -- 'bash tools/experiments/regression_ab.sh'          -> gpurun_out/ab/regression_ab.txt
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $OUT
LOG=$OUT/regression_ab.txt; : > $LOG
STEPS=${STEPS:-20}
want="--steps $STEPS --warmup 3 --cpu-sample 0 --no-node --no-e2e --no-withdraw --no-poseidon --no-sweep --no-deep-state --no-export"
line='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print("%.0f tx/s  %.3f ms/step  smt %.2f front %.2f hash4 %.2f eddsa %.2f fix %.2f hi %.2f%s" % (d["value"], d["ms_per_step"], k.get("smt",0), k.get("front",0), k.get("hash4",0), k.get("eddsa",0), k.get("eddsa_fix",0), k.get("hash_inputs",0), (" fee_acc %.2f" % k["fee_acc"]) if "fee_acc" in k else ""))'
run() {   # run <label> <dir> [env / extra args...]
  local label=$1 dir=$2; shift 2
  local envs="" extra=""
  for a in "$@"; do case $a in *=*) envs="$envs $a";; *) extra="$extra $a";; esac; done
  local args=""
  for f in $want; do   # only the flags this tree's bench.py knows
    case $f in --*) grep -q -- "\"$f\"" $dir/bench.py && keep=1 || keep=0; [ $keep = 1 ] && args="$args $f";; *) [ $keep = 1 ] && args="$args $f";; esac
  done
  local res=$(cd $dir && env $envs timeout 400 python bench.py $args $extra 2>$OUT/last_err.log | grep '^{' | tail -1 | python -c "$line" 2>&1 | tail -1)
  echo "$label: $res" | tee -a $LOG
}
echo "# $(date -u +%FT%TZ)  $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Series' | sed 's/.*: *//')  bench.py $want" | tee -a $LOG
case ${1:-all} in
all|ab)
  for i in 1 2 3; do
    run "r4    (e8a27d7) #$i" variants/ab/r4
    run "head5 (23aaee9) #$i" variants/ab/head5
  done ;;&
all|bisect)
  for i in 1 2; do
    run "a e98be96 k_smt without scratch      #$i" variants/ab/a_e98be96
    run "b bc0ddb2 before k_main_feeacc       #$i" variants/ab/b_bc0ddb2
    run "c 7d8a146 k_main_feeacc              #$i" variants/ab/c_7d8a146
    run "d 67cadea before the split prologue  #$i" variants/ab/d_67cadea
    run "e 07fcb01 split prologue             #$i" variants/ab/e_07fcb01
  done ;;&
all|now)
  for i in 1 2; do
    run "this tree, marks + rotation          #$i" .
    run "this tree, HZ_NO_ZMARK=1 + rotation  #$i" . HZ_NO_ZMARK=1
    run "this tree, marks, --no-rotate        #$i" . --no-rotate
    run "this tree, HZ_NO_ZMARK=1 --no-rotate #$i" . HZ_NO_ZMARK=1 --no-rotate
  done ;;
esac
cat $OUT/last_err.log 2>/dev/null | tail -5
