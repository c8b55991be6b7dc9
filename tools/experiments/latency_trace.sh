# kernel trace of single-batch steps (latency scheduling) -> list of the kernels of the last step with start / end relative to its first kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lat; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-poseidon --no-withdraw --no-e2e --no-shard --distinct-batches 1 --inflight 1 --batches-per-launch 1 "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/last_step.txt <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("hz::", "").replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))))
# steps start with k_main_front; take the 3rd from the end (timed-region steps; later ones are latency / profiling passes)
fronts = [i for i, r in enumerate(rows) if r[2].startswith("k_main_front")]
for which in (2, len(fronts) // 2):
    i0 = fronts[which]
    i1 = fronts[which + 1] if which + 1 < len(fronts) else len(rows)
    t0 = rows[i0][0]
    print("---- step starting at row", i0)
    for s, e, n in rows[max(0, i0 - 6):i1]:
        print("%9.3f %9.3f %8.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n[:40]))
PY
cat $OUT/last_step.txt | head -120
find $OUT -name "*.csv" -delete
