python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu_e.log 2>&1; tail -6 gpurun_out/r03_pytest_gpu_e.log
python bench.py --no-cpu-baseline > gpurun_out/r03_bench_e.json 2> gpurun_out/r03_bench_e.err; python tools/show_bench.py gpurun_out/r03_bench_e.json | cut -c1-400
python bench.py --no-cpu-baseline --no-fp64-line --storage fp64 > gpurun_out/r03_bench_e_fp64.json 2> gpurun_out/r03_bench_e_fp64.err; python tools/show_bench.py gpurun_out/r03_bench_e_fp64.json | cut -c1-400
FSI_GCR_NC64=8 python bench.py --no-cpu-baseline --no-fp64-line --storage fp64 > gpurun_out/r03_bench_e_fp64_nc8.json 2> /dev/null; python tools/show_bench.py gpurun_out/r03_bench_e_fp64_nc8.json | cut -c1-400
