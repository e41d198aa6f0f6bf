#!/bin/bash
# two-level displacement solve: smoothing / coarse-sweep settings -> cylinder golden errors + Krylov counts, then the 1M bench
mkdir -p gpurun_out
i=0
while read -r cfg; do
  [ -z "$cfg" ] && continue
  i=$((i+1))
  echo "== $cfg"
  env $cfg timeout -k 10 120 python tools/mg_check.py 1 2>&1 | grep -a "^mg" | cut -c1-330
  env $cfg timeout -k 10 240 python bench.py --no-cpu-baseline > gpurun_out/mg_$i.json 2> gpurun_out/mg_$i.err
  python tools/show_bench.py gpurun_out/mg_$i.json | cut -c1-330
done
