#!/bin/bash
# Chebyshev smoothers of the 4th kind in the two-level cycles (FSI_CHEB4 bit 0 solid, bit 1 displacement) against the tuned
# 1st-kind intervals: bench command, Krylov iterations / ms per application / ms per step.
O=gpurun_out/cheb4
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-fp64-line > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$tag".ljust(28), "$*".ljust(60), "krylov", d["krylov_iterations"], "ms/app %.2f" % (d["phase_ms"]["precond_ms"]/d["phase_calls"]["precond_calls"]), "ms/step %.1f" % d["ms_per_step"], d["solver_events"])
PY
}
run base_a FSI_CHEB4=0
run s4_16_a FSI_CHEB4=1
run base_b FSI_CHEB4=0
run s4_16_b FSI_CHEB4=1
run s4_14 FSI_CHEB4=1 FSI_SBMG_PRE=14 FSI_SBMG_POST=14
run s4_16_12 FSI_CHEB4=1 FSI_SBMG_PRE=16 FSI_SBMG_POST=12
