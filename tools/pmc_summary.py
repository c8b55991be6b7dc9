"""Summaries of the rocprofv3 --pmc passes tools/gpu_suite.sh takes (committed under profiles/ as rNN_valu_counters.* / rNN_poseidon_valu.json;
bench.py prices kernels against the integer-issue roofline with them). python tools/pmc_summary.py valu|poseidon <bench arguments / command>"""
import sys

mode = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]


def valu():

    import csv, glob, collections, os
    out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc"
    for f in glob.glob(out + "/valu/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); n[k] += 1
        with open(out + "/valu_summary.csv", "w") as o:
            names = sorted({c for k in acc for c in acc[k]})
            o.write("kernel,dispatches," + ",".join(names) + "\n")
            for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0)):
                o.write(k + "," + str(n[k]) + "," + ",".join("%.4g" % (acc[k][c] / n[k]) for c in names) + "\n")
        print(open(out + "/valu_summary.csv").read())
        # per step / per transaction-grid launch, for bench.py's roofline_valu: a step = one k_main_front dispatch; the transaction launch of
        # a kernel = its dispatches with the most waves
        import json
        per = collections.defaultdict(list)   # kernel -> [(SQ_WAVES, SQ_INSTS_VALU)] per dispatch
        cur = {}
        for r in csv.DictReader(open(f)):
            k, d = r["Kernel_Name"].split("(")[0], r["Dispatch_Id"]
            e = cur.setdefault(d, {"k": k, "w": 0.0, "v": 0.0})
            if r["Counter_Name"] == "SQ_WAVES": e["w"] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU": e["v"] += float(r["Counter_Value"])
        for e in cur.values():
            per[e["k"]].append((e["w"], e["v"]))
        steps = max(1, len(per.get("hz::k_main_front", [])))
        kernels = {}
        for k, v in per.items():
            wmax = max(x[0] for x in v)
            big = [x[1] for x in v if x[0] >= 0.9 * wmax]
            key = "k_" + k.replace("void ", "").replace("hz::", "").split("<")[0].replace("k_", "")
            e = kernels.setdefault(key, {"insts_valu_per_step": 0.0, "insts_valu_largest_grid_mean": 0.0, "dispatches": 0})
            e["insts_valu_per_step"] += sum(x[1] for x in v) / steps
            e["insts_valu_largest_grid_mean"] = max(e["insts_valu_largest_grid_mean"], sum(big) / len(big))
            e["dispatches"] += len(v)
        json.dump({"command": "python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-sweep " + " ".join(__import__("sys").argv[1:]), "steps_enqueued": steps,
                   "insts_valu_per_step": sum(e["insts_valu_per_step"] for e in kernels.values()),
                   "note": "SQ_INSTS_VALU (wave-instructions), rocprofv3 --pmc pass of its own; per step = all dispatches / k_main_front dispatches; largest_grid_mean = mean over the dispatches with the most waves (the transaction launch)",
                   "kernels": kernels}, open(out + "/valu_counters.json", "w"), indent=1)
        print("insts_valu_per_step %.4g" % sum(e["insts_valu_per_step"] for e in kernels.values()))


def poseidon():

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


{"valu": valu, "poseidon": poseidon}[mode]()
