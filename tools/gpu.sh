#!/bin/bash
# ONE parameterised script for every gpurun call (round 5; replaces the one-shot tools/gpu_r4_{a..z}.sh):
#
#     gpurun --timeout 1100 -- 'bash tools/gpu.sh <tag> <step> [<step> ...]'
#
# Steps run in order; output goes to gpurun_out/<tag>/.  After a step that was killed at its limit nothing else is started.
#   info                      host cores / memory / GPU
#   tests[:<-k expr>]         pytest -m gpu (optionally a -k selection)          -> pytest_gpu.log
#   smoke                     __graft_entry__.smoke()                            -> smoke.log
#   bench:<label>:<args>      python bench.py <args> (args with ',' for ' ')    -> bench_<label>.json
#   env:<NAME=VALUE>          export for the steps that follow (unset with env:NAME=)
#   trace:<label>:<args>      rocprofv3 --kernel-trace --stats of bench.py <args> -> kernel_stats_<label>.csv, trace_<label>.json
#   pmc:<label>:<args>        FETCH_SIZE / WRITE_SIZE passes (own runs, --kernel-trace only) of bench.py --warmup 0 <args>,
#                             calibrated on the known-byte streams                 -> pmc_traffic_<label>.json
#   py:<label>:<script,args>  python <script> <args>                              -> <label>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
say() { echo "[gpu.sh $(date +%H:%M:%S)] $*"; }
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  case $kind in
    info)
      { nproc; free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $O/info.txt 2>&1; cat $O/info.txt | head -6 ;;
    env)
      if [ "${rest#*=}" = "" ]; then unset "${rest%%=*}"; say "unset ${rest%%=*}"; else export "$rest"; say "export $rest"; fi ;;
    tests)
      if [ -n "$rest" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -q -rs -k "$rest" > $O/pytest_gpu.log 2>&1
      else timeout -k 10 1100 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; fi
      rc=$?; say "pytest rc=$rc"; tail -8 $O/pytest_gpu.log; if [ $rc -eq 124 ]; then exit 1; fi ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; say "smoke rc=$?"; tail -2 $O/smoke.log ;;
    bench)
      label=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""; args=${args//,/ }
      t0=$(date +%s)
      timeout -k 10 1000 python bench.py $args > $O/bench_$label.json 2> $O/bench_$label.err
      rc=$?; say "bench $label ($args) rc=$rc wall $(( $(date +%s) - t0 )) s"; python tools/show_bench.py $O/bench_$label.json | cut -c1-600
      if [ $rc -ne 0 ]; then tail -5 $O/bench_$label.err; fi; if [ $rc -eq 124 ]; then exit 1; fi ;;
    trace)
      label=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""; args=${args//,/ }
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$label &&
        timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$label -- python3 $R/bench.py $args > $O/trace_$label.json 2> $O/trace_$label.err )
      rc=$?; say "trace $label rc=$rc"
      find /tmp/kt_$label -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$label.csv
      if [ -n "$KEEP_TRACE" ]; then find /tmp/kt_$label -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_$label.csv; fi
      rm -rf /tmp/kt_$label; head -14 $O/kernel_stats_$label.csv | cut -c1-160; if [ $rc -eq 124 ]; then exit 1; fi ;;
    pmc)
      label=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""; args=${args//,/ }
      P=$O/pmc_$label; mkdir -p $P
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp &&
          PMC_STEPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmccal_$c -- python3 $R/tools/pmc_driver.py > $P/driver_$c.log 2> $P/driver_$c.err )
        rc=$?; say "pmc calibration $c rc=$rc"; if [ $rc -eq 124 ]; then exit 1; fi
        python3 tools/pmc_summary.py "/tmp/pmccal_$c/**/*counter_collection*.csv" > $P/cal_${c}_summary.csv
        ( cd /tmp && export TMPDIR=/tmp &&
          timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcbench_$c -- python3 $R/bench.py --warmup 0 --no-cpu-baseline --no-side-line $args > $P/bench_$c.json 2> $P/bench_$c.err )
        rc=$?; say "pmc bench $c rc=$rc"; if [ $rc -eq 124 ]; then exit 1; fi
        python3 tools/pmc_summary.py "/tmp/pmcbench_$c/**/*counter_collection*.csv" > $P/bench_${c}_summary.csv
        rm -rf /tmp/pmccal_$c /tmp/pmcbench_$c
      done
      python3 tools/pmc_traffic.py $P/bench_FETCH_SIZE_summary.csv $P/bench_WRITE_SIZE_summary.csv 4294967296 \
              $P/cal_FETCH_SIZE_summary.csv $P/cal_WRITE_SIZE_summary.csv $P/bench_FETCH_SIZE.json > $O/pmc_traffic_$label.json
      python3 tools/pmc_traffic.py --show $O/pmc_traffic_$label.json ;;
    py)
      label=${rest%%:*}; args=${rest#*:}; args=${args//,/ }
      timeout -k 10 900 python $args > $O/$label.txt 2> $O/$label.err; rc=$?; say "py $label rc=$rc"; tail -25 $O/$label.txt | cut -c1-250
      if [ $rc -ne 0 ]; then tail -5 $O/$label.err; fi; if [ $rc -eq 124 ]; then exit 1; fi ;;
    *) say "unknown step $step"; exit 2 ;;
  esac
done
exit 0
