#!/bin/bash
# HBM-side counters of the bench kernels (separate passes, kernel-trace only)
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_$c.json 2> $R/gpurun_out/prof/pmc_$c.err; echo "pmc $c rc=$?"
  python3 $R/tools/pmc_summary.py "/tmp/pmc_$c/**/*counter_collection*.csv" > $R/gpurun_out/prof/pmc_${c}_summary.csv
  head -8 $R/gpurun_out/prof/pmc_${c}_summary.csv
done
