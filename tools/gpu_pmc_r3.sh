#!/bin/bash
# HBM-side counters of the solver kernels with a known-byte calibration (separate passes, kernel-trace only; never combined
# with any other trace domain):
#   pass A: tools/pmc_driver.py (calibration streams + 1 step)  -> calibration factors of FETCH_SIZE / WRITE_SIZE
#   pass B: the bench command itself (python3 bench.py --no-cpu-baseline --no-fp64-line) -> per-kernel mean bytes per launch
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  PMC_STEPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmccal_$c -- python3 $R/tools/pmc_driver.py > $R/gpurun_out/pmc/driver_$c.log 2> $R/gpurun_out/pmc/driver_$c.err; echo "pmc cal $c rc=$?"
  python3 $R/tools/pmc_summary.py "/tmp/pmccal_$c/**/*counter_collection*.csv" > $R/gpurun_out/pmc/cal_${c}_summary.csv
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcbench_$c -- python3 $R/bench.py --no-cpu-baseline --no-fp64-line > $R/gpurun_out/pmc/bench_$c.json 2> $R/gpurun_out/pmc/bench_$c.err; echo "pmc bench $c rc=$?"
  python3 $R/tools/pmc_summary.py "/tmp/pmcbench_$c/**/*counter_collection*.csv" > $R/gpurun_out/pmc/bench_${c}_summary.csv
  head -6 $R/gpurun_out/pmc/bench_${c}_summary.csv | cut -c1-160
  rm -rf /tmp/pmccal_$c /tmp/pmcbench_$c
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc/bench_FETCH_SIZE_summary.csv $R/gpurun_out/pmc/bench_WRITE_SIZE_summary.csv 4294967296 \
        $R/gpurun_out/pmc/cal_FETCH_SIZE_summary.csv $R/gpurun_out/pmc/cal_WRITE_SIZE_summary.csv $R/gpurun_out/pmc/bench_FETCH_SIZE.json > $R/gpurun_out/pmc/pmc_traffic.json
python3 $R/tools/show_bench.py $R/gpurun_out/pmc/bench_FETCH_SIZE.json | cut -c1-300
