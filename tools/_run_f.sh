python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_f.log 2>&1; tail -4 gpurun_out/r03_pytest_gpu_f.log
python bench.py --no-cpu-baseline > gpurun_out/r03_bench_f.json 2> gpurun_out/r03_bench_f.err; python tools/show_bench.py gpurun_out/r03_bench_f.json | cut -c1-400
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r03_bench_f.json').read().strip().splitlines()[-1])
print('fp64 storage', j.get('value_fp64_storage'), j.get('ms_per_step_fp64_storage'), j['fp64_storage'].get('phase_ms'))
PY
