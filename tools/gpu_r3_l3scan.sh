#!/bin/bash
# Round 3: dense third level of the solid block (fsi_amg.hip) against the round-2 coarse sweeps, bench workload.
# usage (GPU box): bash tools/gpu_r3_l3scan.sh  -> gpurun_out/r03_l3scan.txt
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r03_l3scan.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" FSI_DEBUG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp64-line > gpurun_out/_l3.json 2> gpurun_out/_l3.err || { echo "FAILED" >> $out; tail -5 gpurun_out/_l3.err >> $out; return 0; }
  grep "level 3" gpurun_out/_l3.err | head -3 >> $out
  python - >> $out <<'PY'
import json
j=json.loads(open('gpurun_out/_l3.json').read().strip().splitlines()[-1])
pm,pc=j['phase_ms'],j['phase_calls']
print(f"  {j['value']:.2f} it/s  {j['ms_per_step']:.1f} ms/step  newton {j['newton_iterations']} krylov {j['krylov_iterations']}  precond {pm['precond_ms']/max(pc['precond_calls'],1):.3f} ms/apply  factor {pm['factor_ms']:.0f} ms  ortho {pm['ortho_ms']:.0f} spmv {pm['spmv_ms']:.0f}")
PY
}
run FSI_SOLID_L3=0
run FSI_SOLID_L3=1
run FSI_L3_CYCLES=2
run FSI_L3_CYCLES=4 FSI_L3_POST=4 FSI_L3_PRE=4
run FSI_L3_DEG=2
run FSI_L3_AGG=96
run FSI_L3_AGG=24
run FSI_L3_ALPHA=60 FSI_L3_POST=8 FSI_L3_PRE=8
cat $out
