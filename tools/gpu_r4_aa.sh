#!/bin/bash
# Round 4, twenty-seventh GPU call: non-temporal loads on the once-read streams of the orthogonalisation (Q) and the flush (Z):
# kernel trace of 10 bench steps.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4aa
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp64-line > $O/trace.json 2> $O/trace.err; echo "trace rc=$?"
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
grep -i "gcr_flush\|gcr_dots<float\|gcr_axpy<float" $O/kernel_stats.csv | cut -c1-220
cd $R; python tools/show_bench.py $O/trace.json | cut -c1-300
