#!/bin/bash
# The bench command over mesh sizes (10 steps after 2 warm-up steps each): Krylov iterations per solve and the solver's
# event counters - a check that no size meets a pathology of the linear solver (round 3: 100 k tets did).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sizes
mkdir -p $O
for t in 50000 100000 200000 400000 700000 2500000; do
  timeout -k 10 500 python bench.py --steps 10 --warmup 2 --tets $t --no-cpu-baseline --no-fp64-line > $O/s$t.json 2> $O/s$t.err; echo "tets $t rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/s$t.json").read().strip().splitlines()[-1])
print(d["config"]["tets"], "tets:", round(d["value"],2), "it/s", round(d["ms_per_step"],1), "ms/step", "newton", d["newton_iterations"], "krylov", d["krylov_iterations"], d["solver_events"])
print("   ", d["krylov_per_solve"])
PY
done
