#!/bin/bash
# solid two-level solve: settings -> cylinder golden errors + Krylov counts, then the 1M bench
mkdir -p gpurun_out
i=0
while read -r cfg; do
  [ -z "$cfg" ] && continue
  i=$((i+1))
  echo "== $cfg"
  env $cfg FSI_DEBUG=1 timeout -k 10 120 python tools/mg_check.py 1 2>&1 | grep -a "^mg\|two-level\|Error" | cut -c1-330 | head -6
  env $cfg timeout -k 10 240 python bench.py --no-cpu-baseline > gpurun_out/sbmg_$i.json 2> gpurun_out/sbmg_$i.err
  python tools/show_bench.py gpurun_out/sbmg_$i.json | cut -c1-330
  grep -a "FsiError" gpurun_out/sbmg_$i.err | tail -1 | cut -c1-200
done
