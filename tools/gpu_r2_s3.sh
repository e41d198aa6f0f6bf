#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s3
FSI_DEBUG_GCR=1 FSI_KRYLOV_FP32=1 timeout -k 10 200 python tools/gpu_debug_gcr.py > gpurun_out/r2s3/dbg_fp32.log 2>&1; echo "dbg fp32 rc=$?"
grep -v "^\[gcr\]" gpurun_out/r2s3/dbg_fp32.log | tail -3 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_r2s3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2s3/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2s3/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r2s3/bench_prof.err | cut -c1-300
python tools/show_bench.py gpurun_out/r2s3/bench_prof.json
f=$(find /tmp/prof_r2s3 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r2s3/kernel_stats.csv; head -22 "$f" | cut -c1-160; fi
