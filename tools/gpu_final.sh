mkdir -p gpurun_out/r3f1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3f1/pytest.log
tail -5 gpurun_out/r3f1/pytest.log
(time timeout 1200 python bench.py) > gpurun_out/r3f1/bench_default.log 2>&1
grep "^{" gpurun_out/r3f1/bench_default.log | tail -1 > gpurun_out/r3f1/bench_line.json
python tools/show_line.py gpurun_out/r3f1/bench_line.json 2>/dev/null || tail -c 1500 gpurun_out/r3f1/bench_line.json
grep -v "^{" gpurun_out/r3f1/bench_default.log | tail -6
