#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 8 gpurun_out/pytest_gpu.log
timeout -k 10 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err; echo "bench rc=$?"
tail -c 2400 gpurun_out/bench_1m.json
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh50k/stenosis.h5', 50000); print(len(m['tets']))
"
timeout -k 10 500 python tools/gpu_run_case.py offset_stenosis /tmp/mesh50k/stenosis.h5 0.001 0.024 > gpurun_out/run50k_25steps.log 2>&1; echo "50k rc=$?"
head -27 gpurun_out/run50k_25steps.log | cut -c1-200
