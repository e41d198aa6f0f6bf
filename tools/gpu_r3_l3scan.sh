#!/bin/bash
# Round 3: dense third level of the solid block (fsi_amg.hip) against the round-2 coarse sweeps, bench workload.
# usage (GPU box): bash tools/gpu_r3_l3scan.sh <tag> "<env assignments>" ...  -> gpurun_out/r03_l3scan_<tag>.txt
set -o pipefail
mkdir -p gpurun_out
tag=$1; shift
out=gpurun_out/r03_l3scan_$tag.txt
: > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  env $cfg FSI_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp64-line > gpurun_out/_l3.json 2> gpurun_out/_l3.err || { echo "FAILED" >> $out; tail -5 gpurun_out/_l3.err >> $out; continue; }
  grep "level 3" gpurun_out/_l3.err | head -2 >> $out
  python - >> $out <<'PY'
import json
j=json.loads(open('gpurun_out/_l3.json').read().strip().splitlines()[-1])
pm,pc=j['phase_ms'],j['phase_calls']
print(f"  {j['value']:.2f} it/s  {j['ms_per_step']:.1f} ms/step  newton {j['newton_iterations']} krylov {j['krylov_iterations']}  precond {pm['precond_ms']/max(pc['precond_calls'],1):.3f} ms/apply  factor {pm['factor_ms']:.0f} ms  ortho {pm['ortho_ms']:.0f} spmv {pm['spmv_ms']:.0f}")
PY
  tail -1 $out
done
