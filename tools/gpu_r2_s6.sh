#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s6
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > gpurun_out/r2s6/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r2s6/pytest.log | tail -12 | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s6/$name.json 2> gpurun_out/r2s6/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s6/$name.json | cut -c1-900; }
run default
run fp32hot FSI_KRYLOV_FP32=1
run cap256 FSI_KRYLOV_CAP=256
run cap192 FSI_KRYLOV_CAP=192
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2s6 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2s6/bench_prof.json 2> $R/gpurun_out/r2s6/bench_prof.err; echo "prof rc=$?"
cd $R
for f in $(find /tmp/prof_r2s6 -name "*kernel_stats*.csv"); do cp $f gpurun_out/r2s6/kernel_stats.csv; done
python tools/show_bench.py gpurun_out/r2s6/bench_prof.json | cut -c1-600
head -24 gpurun_out/r2s6/kernel_stats.csv | cut -c1-170
