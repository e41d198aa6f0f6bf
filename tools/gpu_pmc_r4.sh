#!/bin/bash
# Round 4: HBM-side counters of the bench command on the SAME launch set as its algorithmic bytes.
#   pass A: tools/pmc_driver.py (calibration streams + 1 step)          -> calibration factors of FETCH_SIZE / WRITE_SIZE
#   pass B: python3 bench.py --warmup 0 --steps 25 (no side lines)      -> per-kernel bytes per launch, and - because nothing
#           runs before the timers are reset - a bench line whose launch counts are those of the trace
# Counters in their own passes with --kernel-trace only (never with another trace domain).  Also one plain --kernel-trace
# --stats pass of the default bench command for the per-kernel durations (profiles/r04_final_kernel_stats.csv).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  PMC_STEPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmccal_$c -- python3 $R/tools/pmc_driver.py > $R/gpurun_out/pmc/driver_$c.log 2> $R/gpurun_out/pmc/driver_$c.err; rc=$?; echo "pmc cal $c rc=$rc"
  [ $rc -eq 124 ] && exit 1
  python3 $R/tools/pmc_summary.py "/tmp/pmccal_$c/**/*counter_collection*.csv" > $R/gpurun_out/pmc/cal_${c}_summary.csv
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcbench_$c -- python3 $R/bench.py --warmup 0 --steps 25 --no-cpu-baseline --no-fp64-line > $R/gpurun_out/pmc/bench_$c.json 2> $R/gpurun_out/pmc/bench_$c.err; rc=$?; echo "pmc bench $c rc=$rc"
  [ $rc -eq 124 ] && exit 1
  python3 $R/tools/pmc_summary.py "/tmp/pmcbench_$c/**/*counter_collection*.csv" > $R/gpurun_out/pmc/bench_${c}_summary.csv
  head -6 $R/gpurun_out/pmc/bench_${c}_summary.csv | cut -c1-160
  rm -rf /tmp/pmccal_$c /tmp/pmcbench_$c
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc/bench_FETCH_SIZE_summary.csv $R/gpurun_out/pmc/bench_WRITE_SIZE_summary.csv 4294967296 \
        $R/gpurun_out/pmc/cal_FETCH_SIZE_summary.csv $R/gpurun_out/pmc/cal_WRITE_SIZE_summary.csv $R/gpurun_out/pmc/bench_FETCH_SIZE.json > $R/gpurun_out/pmc/pmc_traffic.json
python3 - <<PY
import json
j=json.load(open("$R/gpurun_out/pmc/pmc_traffic.json"))
for k,g in j.get("groups",{}).items():
    if isinstance(g,dict): print("%-22s launches %6d (trace %6d)  PMC %9.1f MB / launch  algorithmic %9.1f MB  ratio %.2f"%(k,g["launches"],g["trace_launches"],g["pmc_bytes_per_launch"]/1e6,g["algorithmic_bytes_per_launch"]/1e6,g["traffic_over_algorithmic"]))
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline --no-fp64-line > $R/gpurun_out/pmc/bench_trace.json 2> $R/gpurun_out/pmc/bench_trace.err; echo "kernel trace rc=$?"
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/pmc/kernel_stats.csv
rm -rf /tmp/kt
head -12 $R/gpurun_out/pmc/kernel_stats.csv | cut -c1-150
