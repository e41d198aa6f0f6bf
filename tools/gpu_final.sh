#!/bin/bash
# round-end evidence: GPU tests, smoke, the bench line, and the rocprofv3 kernel-trace summary of the same bench command
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
tail -c 3200 gpurun_out/bench_final.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/bench_profiled_final.json 2> $R/gpurun_out/prof/rocprof_final.err; echo "rocprof rc=$?"
for f in $(find /tmp/prof_final -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/prof/kernel_stats_final.csv; done
head -14 $R/gpurun_out/prof/kernel_stats_final.csv | cut -c1-200
