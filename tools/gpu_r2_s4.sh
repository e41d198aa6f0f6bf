#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s4
FSI_DEBUG_GCR=1 FSI_KRYLOV_FP32=1 timeout -k 10 200 python tools/gpu_debug_gcr.py > gpurun_out/r2s4/dbg_fp32.log 2>&1; echo "dbg fp32 rc=$?"
grep -v "^\[gcr\]" gpurun_out/r2s4/dbg_fp32.log | tail -7 | cut -c1-600
FSI_KRYLOV_FP32=0 timeout -k 10 200 python tools/gpu_debug_gcr.py > gpurun_out/r2s4/dbg_fp64.log 2>&1; echo "dbg fp64 rc=$?"
grep -v "^\[gcr\]" gpurun_out/r2s4/dbg_fp64.log | tail -7 | cut -c1-600
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2s4/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r2s4/pytest.log | cut -c1-300
run() { name=$1; shift; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2s4/$name.json 2> gpurun_out/r2s4/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s4/$name.json | cut -c1-700; }
run fp32 FSI_KRYLOV_FP32=1
run fp64 FSI_KRYLOV_FP32=0
run fp32_f1e-2 FSI_KRYLOV_FP32=1 FSI_NEWTON_FORCING=1e-2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2s4 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r2s4/bench_prof.json 2> $R/gpurun_out/r2s4/bench_prof.err; echo "prof rc=$?"
cd $R
for f in $(find /tmp/prof_r2s4 -name "*kernel_stats*.csv"); do cp $f gpurun_out/r2s4/kernel_stats.csv; done
head -20 gpurun_out/r2s4/kernel_stats.csv | cut -c1-170
