#!/bin/bash
# Final measurement of round 3, part A (one gpurun call): GPU tests, smoke, the bench line (with value_fp64_storage and the
# one-lifetime CPU baseline; wall time printed: the driver allows 600 s), the same command under rocprofv3 --stats.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "FAILED|ERROR|passed|failed" $O/pytest.log | tail -4 | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
T0=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python tools/show_kernels.py $O/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 $R/bench.py --no-cpu-baseline --no-fp64-line > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "prof rc=$?"
cd $R
for f in $(find /tmp/prof_final -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats.csv; done
python tools/show_kernels.py $O/bench_profiled.json | head -1 | cut -c1-200
