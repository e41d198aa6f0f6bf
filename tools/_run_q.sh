python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest_gpu_q.log 2>&1; tail -3 gpurun_out/r03_pytest_gpu_q.log
python bench.py --no-cpu-baseline > gpurun_out/r03_bench_q.json 2>gpurun_out/r03_bench_q.err
python tools/show_bench.py gpurun_out/r03_bench_q.json | cut -c1-330
python -c "
import json
d=json.loads(open('gpurun_out/r03_bench_q.json').read().strip().splitlines()[-1]); print('fp64 storage', d.get('value_fp64_storage'), d['solver_events'])"
FSI_SPMV_SIDE=0 python bench.py --no-cpu-baseline --no-fp64-line > gpurun_out/r03_bench_q0.json 2>gpurun_out/r03_bench_q0.err
python tools/show_bench.py gpurun_out/r03_bench_q0.json | cut -c1-330
