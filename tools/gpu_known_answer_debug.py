"""The five known-answer steps of the offset-stenosis fixture (dt = 0.01, theta = 0.51, one Jacobian for all steps) through the HIP
backend with named FsiTuning overrides: Newton histories (|b|, |du|, Krylov iterations, achieved relres) per step.
    python tools/gpu_known_answer_debug.py [name=value ...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from conftest import GOLDEN, prepare_case
from vasp_amd.capi import HipBackend
import tempfile, contextlib, io
tune = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    tune[k] = float(v) if "." in v or "e" in v else int(v)
case = prepare_case("offset_stenosis", GOLDEN / "offset_stenosis" / "offset_stenosis.h5", tempfile.mkdtemp(), dt="0.01", T="0.04")
ns, desc, bc_values, pressure, hook = case
hb = HipBackend(desc, tuning=tune or None)
print("tuning overrides:", tune)
for k in range(5):
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 0.01 * (k + 1); hook("pre_solve")(**ns)
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
    h = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                        recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
    hb.shift()
    print("step", k, len(h), "its:", " ".join(f"[|b| {x[0]:.2e} |du| {x[1]:.2e} k{x[3]} rr {x[4]:.1e}]" for x in h))
U = hb.get_state("n")
gold = np.load(GOLDEN / "stenosis_tight.npz")["states"][4]
N2 = ns["mesh"].num_nodes
for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
    print(name, "distance to the converged oracle run:", np.linalg.norm(U[sl] - gold[sl]) / np.linalg.norm(gold[sl]))
hb.close()
