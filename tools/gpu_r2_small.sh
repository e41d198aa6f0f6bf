#!/bin/bash
# how busy is the card at a partition-sized problem (140 k tets = the bench mesh over 8 ranks)?
mkdir -p gpurun_out/small
R=$PWD
timeout -k 10 300 python bench.py --tets 140000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/small/bench_140k.json 2> gpurun_out/small/bench_140k.err; echo "bench rc=$?"
python tools/show_bench.py gpurun_out/small/bench_140k.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/small/prof -o k140 -- python3 $R/bench.py --tets 140000 --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/small/prof.json 2> $R/gpurun_out/small/prof.err; echo "prof rc=$?"
cd $R
find gpurun_out/small/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/small/kernel_stats.csv
find gpurun_out/small/prof -name "*.db" -delete; find gpurun_out/small/prof -name "*kernel_trace.csv" -delete
python tools/show_bench.py gpurun_out/small/prof.json | cut -c1-600
