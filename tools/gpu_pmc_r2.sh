#!/bin/bash
# HBM-side counters of the solver kernels with a known-byte calibration (separate passes, kernel-trace only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/pmc_driver.py > $R/gpurun_out/pmc/driver_$c.log 2> $R/gpurun_out/pmc/driver_$c.err; echo "pmc $c rc=$?"
  tail -3 $R/gpurun_out/pmc/driver_$c.log
  python3 $R/tools/pmc_summary.py "/tmp/pmc_$c/**/*counter_collection*.csv" > $R/gpurun_out/pmc/pmc_${c}_summary.csv
  head -8 $R/gpurun_out/pmc/pmc_${c}_summary.csv | cut -c1-160
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc/pmc_FETCH_SIZE_summary.csv $R/gpurun_out/pmc/pmc_WRITE_SIZE_summary.csv 4294967296 > $R/gpurun_out/pmc/pmc_traffic.json
head -20 $R/gpurun_out/pmc/pmc_traffic.json
