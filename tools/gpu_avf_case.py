"""BASELINE config 4's problem file (avf: two MooneyRivlin regions, Robin on both outer walls, pulsatile table inflow,
dt = 1e-4) on a synthetic tube of N tets, K steps through the HIP backend: Newton / Krylov counts and solver events.
usage: python tools/gpu_avf_case.py N K   (GPU box helper; the mesh is the one of tests/conftest.make_avf_case at size N)"""
import sys, json, time, contextlib, io, tempfile, pathlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from conftest import prepare_case
from vasp_amd.mesh import FsiMesh
from vasp_amd.meshgen import generate
from vasp_amd.capi import HipBackend

N, K = int(sys.argv[1]), int(sys.argv[2])
tmp = pathlib.Path(tempfile.mkdtemp())
m = generate(N)
x_c = m["coords"][m["tets"]].mean(axis=1)[:, 0]
x_f = m["coords"][m["facets"]].mean(axis=1)[:, 0]
cm, fm = m["cell_markers"].copy(), m["facet_markers"].copy()
mid = 0.008
cm[(cm == 2) & (x_c > mid)] = 1002
for a, b in ((11, 1011), (22, 1022), (33, 1033)):
    fm[(fm == a) & (x_f > mid)] = b
FsiMesh.from_arrays(m["coords"], m["tets"], cm, m["facets"], fm).write(tmp / "avf.h5")
(tmp / "avf_probe_point.json").write_text(json.dumps([[0.0, 0.0, 0.0], [16.0, 0.0, 0.0]]))
(tmp / "avf.csv").write_text("v_PA,v_DA,PV\n" + "\n".join(f"{0.3 + 0.01 * i},{0.1 + 0.005 * i},{9000 + 50 * i}" for i in range(20)))
case = prepare_case("avf", tmp / "avf.h5", tmp / "run", dt="0.0001", T="0.2", theta="0.501",
                    extra=(f"patient_data_path={tmp / 'avf.csv'}", "fsi_region=[0.008,0,0,0.006]"))
ns, desc, bc_values, pressure, hook = case
hb = HipBackend(desc)
print("tets", len(m["tets"]), "dofs", hb.ndof, flush=True)
t0 = time.time()
tot_n = tot_k = 0
for k in range(K):
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 1e-4 * (k + 1); hook("pre_solve")(**ns)
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
    hist = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns.get("max_it", 50), lmbda=1.0,
                           recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
    hb.shift()
    tot_n += len(hist); tot_k += sum(h[3] for h in hist)
    print("step", k, [(f"{h[0]:.1e}", f"{h[1]:.1e}", int(h[2]), h[3]) for h in hist], flush=True)
tm = hb.timers()
print("newton", tot_n, "krylov", tot_k, "seconds", round(time.time() - t0, 2),
      {k: int(tm[k]) for k in ("gcr_arnoldi_steps", "gcr_restarts", "newton_retries", "fp32_fallbacks")}, "q bytes", tm["q_elem_bytes"])
hb.close()
