mkdir -p gpurun_out/r3nb
timeout 900 python -m pytest tests/test_native_builder.py tests/test_batch_builder_device.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3nb/pytest.log
tail -5 gpurun_out/r3nb/pytest.log
(time timeout 1200 python bench.py) > gpurun_out/r3nb/bench_default.log 2>&1
grep "^{" gpurun_out/r3nb/bench_default.log | tail -1 > gpurun_out/r3nb/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3nb/bench_line.json'))
print(d['value'], d['ms_per_step'], d.get('value_e2e'), d.get('value_node'))
print(d['config'].get('batch_builder'), d['config'].get('batch_build_s'))
print(d.get('deep_state',{}).get('value'), d.get('deep_state',{}).get('batch_build_s'), d.get('deep_state',{}).get('state_build_s'))
PY
grep -v "^{" gpurun_out/r3nb/bench_default.log | tail -6
