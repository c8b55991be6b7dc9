#!/bin/bash
# Everything that is committed under profiles/ for a round, in one gpurun call: the default bench line, rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE / VALU counter passes of the same command (separate --pmc passes), a kernel Gantt of the timed region, the
# microbenchmarks behind DESIGN.md 3 and the overlap / power probes. Outputs under gpurun_out/round/.
cd /tmp && export TMPDIR=/tmp
export HZ_ROUND=6
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round; rm -rf $OUT; mkdir -p $OUT
cd $R
( time timeout 900 python bench.py ) > $OUT/bench_default.log 2>&1
grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_line.json
bash tools/profile.sh > $OUT/profile.log 2>&1
cp gpurun_out/prof/kernel_stats.csv gpurun_out/prof/hbm_counters.csv gpurun_out/prof/hbm_counters.json $OUT/ 2>/dev/null
bash tools/gpu_suite.sh pmc --no-e2e --no-deep-state --no-withdraw --no-poseidon --distinct-batches 4 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc/valu_summary.csv $OUT/valu_counters.csv 2>/dev/null; cp gpurun_out/pmc/valu_counters.json $OUT/valu_counters.json 2>/dev/null
# the same pass on a DEEP state (2^20 accounts: every level of every proof hashes): deep_state.k_smt.frac_valu of the bench line
PMC_TIMEOUT=400 bash tools/gpu_suite.sh pmc --no-e2e --no-deep-state --no-withdraw --no-poseidon --distinct-batches 4 --accounts 1048576 > $OUT/pmc_deep.log 2>&1
cp gpurun_out/pmc/valu_counters.json $OUT/valu_counters_deep.json 2>/dev/null
python tools/resource_usage.py > $OUT/resource_usage.txt 2>&1
TL_OUT=round_tl bash tools/gpu_suite.sh timeline --no-withdraw --no-e2e --no-deep-state --no-sweep --distinct-batches 4 > $OUT/timeline.log 2>&1
f=$(find gpurun_out/round_tl -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  python - "$f" > $OUT/timeline_gantt.txt <<'PY'
import csv, subprocess, sys
rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
t0 = rows[0][0]
fr = [(s - t0) / 1e6 for s, n in rows if "k_main_front" in n]
lo = fr[4] - 5 if len(fr) > 8 else fr[0]
print("# kernel Gantt of two steady-state steps of the default bench (2 ms per column; digit = launches of that kernel running)")
sys.stdout.flush()
subprocess.run([sys.executable, "tools/gantt.py", sys.argv[1], str(lo), str(lo + 230), "2"])
PY
fi
if [ -z "$SKIP_MICRO" ]; then   # SKIP_MICRO=1: only what depends on the library (bench line, kernel stats, counters, Gantt)
# every probe under its own timeout: a missing binary or a stuck sampler must not eat the call (round 4 lost 50 GPU-minutes to one)
for b in instbench mulbench invbench mfmabench storebench; do [ -x tools/microbench/$b ] || echo "tools/microbench/$b not built (tools/microbench/build.sh)"; done
timeout 120 tools/microbench/instbench > $OUT/instbench.txt 2>&1
timeout 300 tools/microbench/mulbench > $OUT/mulbench.txt 2>&1
timeout 120 tools/microbench/invbench > $OUT/invbench.txt 2>&1
timeout 180 python tools/microbench/run_mfmabench.py > $OUT/mfmabench.txt 2>&1
( timeout 120 tools/microbench/storebench 1048576 256 5; timeout 120 tools/microbench/storebench 65536 2048 5 ) > $OUT/storebench.txt 2>&1
( timeout 300 python tools/experiments/overlap_probe.py 3; timeout 300 python tools/experiments/overlap_probe.py 5 ) > $OUT/overlap_probe.txt 2>/dev/null
timeout 120 python tools/experiments/power_probe.py 2>/dev/null | python -c "
import sys,re
for l in sys.stdin:
    l=l.rstrip()
    m=re.search(r'sclk clock speed:.*\|\s*card0,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\d+,\((\d+)Mhz\),S,([0-9.]+)', l)
    print('    fclk %s mclk %s sclk %s MHz, package power %s W' % (m.group(1), m.group(2), m.group(3), m.group(5)) if m else l)
" > $OUT/power_probe.txt
fi
rm -rf gpurun_out/round_tl gpurun_out/prof/trace gpurun_out/prof/pmc_fetch gpurun_out/prof/pmc_write gpurun_out/pmc/valu
tail -c 600 $OUT/bench_line.json; echo; ls -la $OUT
