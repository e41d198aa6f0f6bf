#!/bin/bash
mkdir -p gpurun_out/prof
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err; echo "bench rc=$?"
tail -c 2600 gpurun_out/bench_1m.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof/rocprof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find /tmp/prof_r1 -name "*stats*" | head; for f in $(find /tmp/prof_r1 -name "*kernel_stats*.csv"); do cp $f gpurun_out/prof/; done
ls -la gpurun_out/prof; head -25 gpurun_out/prof/*kernel_stats*.csv
