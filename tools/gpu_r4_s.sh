#!/bin/bash
# Round 4, nineteenth GPU call: the solid cycle's coarse interval following the size of its level (sbmg_ckappa_per_node 0.18,
# against 0 = the fixed 4000 / 90): bench workload at four sizes, two ranks on one card, the aneurysm and avf problem files.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4s
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
    pm=d["phase_ms"]; pc=d["phase_calls"]; k=max(1,pc["precond_calls"])
    print("%-22s %7.2f it/s %6.1f ms/step newton %3d krylov %4d precond %.3f ortho %.3f spmv %.3f ms/it ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/k, pm["ortho_ms"]/k, pm["spmv_ms"]/k, {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run t48_rule     48000 20 5 A=1
run s140_rule    140000 20 5 A=1
run u346_rule    346000 20 5 A=1
run m1_rule      1000000 20 5 A=1
for v in 0 0.18; do
  FSI_SBMG_CKAPPA_PER_NODE=$v VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 8 --warmup 2 --tets 280000 --no-cpu-baseline > $O/two_$v.json 2> $O/two_$v.err
  echo "2 ranks, 280 k tets, per_node $v rc=$?"; python tools/show_bench.py $O/two_$v.json | cut -c1-260
done
for v in 0 0.18; do
  FSI_SBMG_CKAPPA_PER_NODE=$v timeout -k 10 300 python tools/gpu_aneurysm_case.py 346000 10 > $O/aneurysm346_$v.txt 2>&1; echo "aneurysm 346k per_node $v rc=$?"; tail -1 $O/aneurysm346_$v.txt | cut -c1-200
  FSI_SBMG_CKAPPA_PER_NODE=$v timeout -k 10 300 python tools/gpu_avf_case.py 48000 25 > $O/avf48_$v.txt 2>&1; echo "avf 48k per_node $v rc=$?"; tail -1 $O/avf48_$v.txt | cut -c1-200
  FSI_SBMG_CKAPPA_PER_NODE=$v timeout -k 10 300 python tools/gpu_avf_case.py 346000 12 > $O/avf346_$v.txt 2>&1; echo "avf 346k per_node $v rc=$?"; tail -1 $O/avf346_$v.txt | cut -c1-200
done
