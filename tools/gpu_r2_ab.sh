#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/ab/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/ab/$name.json | cut -c1-200; }
run a
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/ab/bench_prof.json 2> $R/gpurun_out/ab/bench_prof.err; echo "prof rc=$?"
cd $R
for f in $(find /tmp/prof_ab -name "*kernel_stats*.csv"); do cp $f gpurun_out/ab/kernel_stats.csv; done
