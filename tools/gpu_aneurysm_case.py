"""BASELINE config 3 / 5's problem file (aneurysm: StVK wall, Robin condition k_s = 1e5, c_s = 10 on the outer wall,
Womersley inlet, ramped interface pressure; its own atol = 1e-10, rtol = 1e-9) on the synthetic tube of N tets, K steps
through the HIP backend: Newton / Krylov counts, solver events, the basis storage the policy chose.
usage: python tools/gpu_aneurysm_case.py N K [name=value ...]   (GPU box helper; mesh of vasp_amd.meshgen at size N; further
arguments go to the problem file as --new-arguments, e.g. atol=1e-6 rtol=1e-6 for the mixed-storage path of config 5)"""
import sys, json, time, contextlib, io, tempfile, pathlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import prepare_case
from vasp_amd.mesh import FsiMesh
from vasp_amd.meshgen import generate
from vasp_amd.capi import HipBackend

N, K = int(sys.argv[1]), int(sys.argv[2])
tmp = pathlib.Path(tempfile.mkdtemp())
m = generate(N)
FsiMesh.from_arrays(m["coords"], m["tets"], m["cell_markers"], m["facets"], m["facet_markers"]).write(tmp / "aneurysm.h5")
(tmp / "aneurysm_probe_point.json").write_text(json.dumps([[0.0, 0.0, 0.0], [16.0, 0.0, 0.0]]))      # mm: scale_probe
case = prepare_case("aneurysm", tmp / "aneurysm.h5", tmp / "run", dt="0.001", T="0.2", theta="0.501", extra=tuple(sys.argv[3:]))
ns, desc, bc_values, pressure, hook = case
hb = HipBackend(desc)
print("tets", len(m["tets"]), "dofs", hb.ndof, "atol", ns["atol"], "rtol", ns["rtol"], flush=True)
tot_n = tot_k = 0
t_steps = []
for k in range(K):
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 1e-3 * (k + 1); hook("pre_solve")(**ns)
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
    t0 = time.time()
    hist = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns.get("max_it", 50), lmbda=1.0,
                           recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
    hb.shift()
    t_steps.append(time.time() - t0)
    tot_n += len(hist); tot_k += sum(h[3] for h in hist)
    print("step", k, "%.3f s" % t_steps[-1], [(f"{h[0]:.1e}", f"{h[1]:.1e}", int(h[2]), h[3]) for h in hist], flush=True)
tm = hb.timers()
print("newton", tot_n, "krylov", tot_k, "seconds", round(sum(t_steps), 2), "Newton-it/s %.2f" % (tot_n / sum(t_steps)),
      {k: int(tm[k]) for k in ("gcr_arnoldi_steps", "gcr_restarts", "newton_retries", "fp32_fallbacks", "verdicts_skipped", "reorth_forced")},
      "q bytes", tm["q_elem_bytes"])
hb.close()
