#!/bin/bash
# SQ_INSTS_VALU of the Poseidon batch kernels (BASELINE's second metric), a --pmc pass of its own: wave-instructions per launch of 2^20
# permutations for each instantiation -> gpurun_out/pmc_pos/poseidon_valu.json (committed as profiles/rNN_poseidon_valu.json; bench.py
# prices the kernels against the integer-issue roofline with it)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_pos; rm -rf $OUT; mkdir -p $OUT
cd $R
CMD="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-sweep --no-e2e --no-deep-state --no-withdraw --no-node --distinct-batches 1 --batches-per-launch 1 --inflight 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/run -o run --output-format csv -- $CMD > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
python - "$CMD" <<'PY'
import csv, glob, collections, json, os, sys
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_pos"
per = {}
for f in glob.glob(out + "/run/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "poseidon_batch_kernel" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hz::", "")
        e = per.setdefault((k, r["Dispatch_Id"]), {"SQ_WAVES": 0.0, "SQ_INSTS_VALU": 0.0})
        if r["Counter_Name"] in e:
            e[r["Counter_Name"]] += float(r["Counter_Value"])
res = {}
wmax = max([e["SQ_WAVES"] for e in per.values()] or [0])   # the launches of 2^20 permutations (two per lane: 8192 wavefronts)
for (k, _), e in per.items():
    if e["SQ_WAVES"] >= wmax:
        res.setdefault(k, []).append(e["SQ_INSTS_VALU"])
js = {"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -- " + sys.argv[1], "permutations_per_launch": 1 << 20,
      "kernels": {k: {"insts_valu_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in sorted(res.items())}}
json.dump(js, open(out + "/poseidon_valu.json", "w"), indent=1)
print(json.dumps(js, indent=1))
PY
find $OUT -name "*.csv" -size +2M -delete
