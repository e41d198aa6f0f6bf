python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest_gpu_q.log 2>&1; tail -3 gpurun_out/r03_pytest_gpu_q.log
python tools/gpu_avf_case.py 50000 25 2>&1 | tail -1 | cut -c1-250
python bench.py --no-cpu-baseline --no-fp64-line > gpurun_out/r03_bench_q.json 2>gpurun_out/r03_bench_q.err
python tools/show_bench.py gpurun_out/r03_bench_q.json | cut -c1-330
FSI_F32_VERDICT_SKIP=1 python bench.py --no-cpu-baseline --no-fp64-line > gpurun_out/r03_bench_q0.json 2>gpurun_out/r03_bench_q0.err
python tools/show_bench.py gpurun_out/r03_bench_q0.json | cut -c1-330
