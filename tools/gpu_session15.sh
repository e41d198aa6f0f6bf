#!/bin/bash
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_profiled.json 2> $R/gpurun_out/prof/rocprof.err; echo "rocprof rc=$?"
for f in $(find /tmp/prof_r1 -name "*kernel_stats*.csv"); do cp $f $R/gpurun_out/prof/kernel_stats.csv; done
head -12 $R/gpurun_out/prof/kernel_stats.csv | cut -c1-220
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_pmc_fetch.json 2> $R/gpurun_out/prof/pmc_fetch.err; echo "pmc fetch rc=$?"
python3 $R/tools/pmc_summary.py "/tmp/pmc_f/**/*counter_collection*.csv" > $R/gpurun_out/prof/pmc_fetch_summary.csv 2>&1; head -8 $R/gpurun_out/prof/pmc_fetch_summary.csv
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_pmc_write.json 2> $R/gpurun_out/prof/pmc_write.err; echo "pmc write rc=$?"
python3 $R/tools/pmc_summary.py "/tmp/pmc_w/**/*counter_collection*.csv" > $R/gpurun_out/prof/pmc_write_summary.csv 2>&1; head -8 $R/gpurun_out/prof/pmc_write_summary.csv
cd $R
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh50k/stenosis.h5', 50000); print(len(m['tets']))
"
timeout -k 10 500 python tools/gpu_run_case.py offset_stenosis /tmp/mesh50k/stenosis.h5 0.001 0.024 > gpurun_out/run50k_25steps.log 2>&1; echo "50k rc=$?"
head -28 gpurun_out/run50k_25steps.log | cut -c1-260
