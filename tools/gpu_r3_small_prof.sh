#!/bin/bash
# is the solver at the per-GPU size of an 8-rank run (140 k tets) bound by the device or by the host's launch rate?
# kernel trace of the bench at that size: sum of kernel durations against the timed region.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/small_prof
python3 bench.py --tets 140400 --steps 10 --warmup 2 --no-cpu-baseline --no-fp64-line > gpurun_out/small_prof/bench_plain.json 2> gpurun_out/small_prof/bench_plain.err
echo plain rc=$?
python tools/show_bench.py gpurun_out/small_prof/bench_plain.json | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -- python3 bench.py --tets 140400 --steps 10 --warmup 2 \
    --no-cpu-baseline --no-fp64-line > gpurun_out/small_prof/bench.json 2> gpurun_out/small_prof/bench.err
echo rc=$?
python tools/show_bench.py gpurun_out/small_prof/bench.json | cut -c1-600
f=$(find /tmp/prof_small -name '*kernel_stats*.csv' | head -1)
cp "$f" gpurun_out/small_prof/kernel_stats.csv
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/small_prof/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernel time total %.1f ms in %d launches (12 steps + setup)'%(tot/1e6,calls))
for r in rows[:12]: print(r['Name'][:60], r['Calls'], '%.1f ms'%(float(r['TotalDurationNs'])/1e6), '%.1f us'%(float(r['AverageNs'])/1e3))
P
