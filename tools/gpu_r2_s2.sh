#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s2
FSI_DEBUG_GCR=1 FSI_KRYLOV_FP32=1 timeout -k 10 300 python tools/gpu_debug_gcr.py > gpurun_out/r2s2/dbg_fp32.log 2>&1; echo "dbg fp32 rc=$?"
FSI_KRYLOV_FP32=0 timeout -k 10 300 python tools/gpu_debug_gcr.py > gpurun_out/r2s2/dbg_fp64.log 2>&1; echo "dbg fp64 rc=$?"
tail -3 gpurun_out/r2s2/dbg_fp32.log; tail -3 gpurun_out/r2s2/dbg_fp64.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2s2/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2s2/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2s2/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/show_bench.py gpurun_out/r2s2/bench_prof.json
f=$(find gpurun_out/r2s2/prof -name "*kernel_stats.csv" | head -1); echo $f; head -25 $f | cut -c1-150
find gpurun_out/r2s2/prof -name "*.db" -delete; find gpurun_out/r2s2/prof -name "*kernel_trace.csv" -size +20M -delete
