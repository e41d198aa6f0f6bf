#!/bin/bash
mkdir -p gpurun_out
for cap in "$@"; do
  FSI_KRYLOV_CAP=$cap timeout -k 10 400 python bench.py --steps 20 --warmup 1 --no-cpu-baseline > gpurun_out/cap_$cap.json 2> gpurun_out/cap_$cap.err
  echo "== cap $cap"; python tools/show_bench.py gpurun_out/cap_$cap.json | cut -c1-420
done
