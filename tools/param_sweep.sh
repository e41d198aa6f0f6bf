#!/bin/bash
# bench.py under different Chebyshev sweep counts / intervals of the block preconditioner (one line per setting)
mkdir -p gpurun_out
i=0
while read -r cfg; do
  [ -z "$cfg" ] && continue
  i=$((i+1))
  env $cfg timeout -k 10 240 python bench.py --no-cpu-baseline > gpurun_out/sweep_$i.json 2> gpurun_out/sweep_$i.err
  echo "== $cfg"
  python tools/show_bench.py gpurun_out/sweep_$i.json | cut -c1-420
done
