#!/bin/bash
# experiment: segment-lane lockstep ladder (HZ_ED_SEG_G): parity under each setting, VALU instruction counts, step times
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/seg; mkdir -p $OUT; rm -f $OUT/seg2.log
for g in 4 3 2; do
  echo "== HZ_ED_SEG_G=$g" | tee -a $OUT/seg2.log
  HZ_ED_SEG_G=$g timeout 1500 python -m pytest tests/test_witness_gpu.py -m gpu -x -q -k "throughput" 2>&1 | tail -3 | tee -a $OUT/seg2.log
done
ARGS="--steps 4 --warmup 1 --cpu-sample 0 --no-e2e --no-deep-state --no-withdraw --no-poseidon --no-node --distinct-batches 4"
cd /tmp && export TMPDIR=/tmp
for g in 0 4; do
  HZ_ED_SEG_G=$g timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $GRAFT_REPO_ROOT/$OUT/valu_g$g -o run -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/valu_g$g.log 2>&1; echo "valu g=$g rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee -a gpurun_out/seg/seg2.log
import csv, glob, collections
for g in (0, 4):
    f = glob.glob("gpurun_out/seg/valu_g%d/**/*counter_collection.csv" % g, recursive=True)
    if not f: print("no counters for", g); continue
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != "SQ_INSTS_VALU": continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hz::", "")
        per[k][0] += 1; per[k][1] += float(r["Counter_Value"])
    print("seg_g=%d  SQ_INSTS_VALU summed over the run (calls, total G):" % g)
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:9]:
        print("   %-40s %5d %9.3f" % (k[:40], v[0], v[1] / 1e9))
PY
find $OUT -name "*.csv" -size +2M -delete
for rep in 1 2; do
for g in 0 4 3; do
  echo "seg_g=$g: $(HZ_ED_SEG_G=$g python bench.py --steps 8 --warmup 3 --cpu-sample 0 --no-withdraw --no-e2e --no-poseidon --no-node --no-deep-state --distinct-batches 8 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernels_ms"])' 2>&1 | tail -1)" | tee -a $OUT/seg2.log
done
done
