python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest_gpu_q.log 2>&1; tail -3 gpurun_out/r03_pytest_gpu_q.log
python tools/gpu_avf_case.py 50000 25 2>&1 | tail -1 | cut -c1-250
python tools/gpu_avf_case.py 300000 12 2>&1 | tail -1 | cut -c1-250
bash tools/gpu_r3_sizes.sh 2>&1 | grep -v "^    " | cut -c1-250
