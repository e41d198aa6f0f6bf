#!/bin/bash
# Round 4, GPU call: non-temporal loads A / B / A / B on one box
# (libvaspfsi_nt0.so = the same tree built with -DFSI_GCR_STREAM_MIB=1e30: never non-temporal).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4ab
mkdir -p $O
cd $R
cp vasp_amd/libvaspfsi.so /tmp/lib_nt1.so; cp vasp_amd/libvaspfsi_nt0.so /tmp/lib_nt0.so
for i in 1 2; do for v in nt0 nt1; do
  cp /tmp/lib_$v.so vasp_amd/libvaspfsi.so
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp64-line > $O/${v}_$i.json 2> $O/${v}_$i.err
  rc=$?
  python - <<PY
import json
d=json.loads([l for l in open("$O/${v}_$i.json") if l.startswith("{")][-1])
pm=d["phase_ms"]; pc=d["phase_calls"]; k=max(1,pc["precond_calls"])
print("%-8s %7.2f it/s %6.1f ms/step krylov %4d precond %.3f ortho %.3f spmv %.3f ms/it roofline frac %.3f" % ("${v}_$i", d["value"], d["ms_per_step"], d["krylov_iterations"], pm["precond_ms"]/k, pm["ortho_ms"]/k, pm["spmv_ms"]/k, d["roofline"]["frac"]))
PY
  [ $rc -eq 124 ] && exit 1
done; done
cp /tmp/lib_nt1.so vasp_amd/libvaspfsi.so
for v in nt0 nt1; do
  cp /tmp/lib_$v.so vasp_amd/libvaspfsi.so
  timeout -k 10 400 python bench.py --steps 12 --warmup 3 --tets 140000 --no-cpu-baseline --no-fp64-line > $O/s140_$v.json 2> $O/s140_$v.err
  python tools/show_bench.py $O/s140_$v.json | cut -c1-330
done
cp /tmp/lib_nt1.so vasp_amd/libvaspfsi.so
