#!/bin/bash
# Round 4, eighth GPU call: tile sizes of the sweep kernels by context size (Schur rows per workgroup 64 / 128 / 256, nodes per
# workgroup 128 / 256) at the per-GPU size of an 8-rank run and at full size; then the GPU test-suite on the chosen defaults.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
run() {   # name tets steps warmup env...
  name=$1; tets=$2; steps=$3; warm=$4; shift 4
  env "$@" timeout -k 10 400 python bench.py --steps $steps --warmup $warm --tets $tets --no-cpu-baseline --no-fp64-line > $O/$name.json 2> $O/$name.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    pm=d["phase_ms"]; pc=d["phase_calls"]; k=d["kernels"]
    sw={n.split(" ")[0]: round(1e3*v["avg_launch_ms"],1) for n,v in k.items() if "sweep" in n or "Schur" in n}
    print("%-24s %8.2f it/s %7.1f ms/step newton %3d krylov %4d precond %.3f ms/app  sweeps [us] %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/max(1,pc["precond_calls"]), sw))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
for st in 64 128 256; do for tn in 128 256; do
  run s140_schur${st}_nodes${tn} 140000 12 3 FSI_SCHUR_TILE=$st FSI_TILE_NODES=$tn
done; done
for st in 128 256; do for tn in 128 256; do
  run m1_schur${st}_nodes${tn} 1000000 20 5 FSI_SCHUR_TILE=$st FSI_TILE_NODES=$tn
done; done
run m1_schur64_nodes256 1000000 20 5 FSI_SCHUR_TILE=64 FSI_TILE_NODES=256
timeout -k 10 1100 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_properties_at_config3_size_with_the_robin_wall > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log
