#!/bin/bash
# HBM-side counters of the bench kernels (separate passes, kernel-trace only) + rehearsal of the partitioned bench
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_$c.json 2> $R/gpurun_out/prof/pmc_$c.err; echo "pmc $c rc=$?"
  python3 $R/tools/pmc_summary.py "/tmp/pmc_$c/**/*counter_collection*.csv" > $R/gpurun_out/prof/pmc_${c}_summary.csv
  head -12 $R/gpurun_out/prof/pmc_${c}_summary.csv
done
cd $R
export VASPFSI_LIN_MAX_IT=600
for n in 2 4; do
  timeout -k 10 300 tools/rehearse_partition.sh $n 1000000 3 > gpurun_out/reh_final_$n.json 2> gpurun_out/reh_final_$n.err; echo "ranks $n rc=$?"
  python tools/show_bench.py gpurun_out/reh_final_$n.json | cut -c1-500
done
timeout -k 10 300 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/single3_final.json 2> /dev/null; python tools/show_bench.py gpurun_out/single3_final.json | cut -c1-300
