"""Cylinder golden comparison with the two-level displacement solve on / off: per-step field errors and Newton histories."""
import os, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import GOLDEN, prepare_case
from test_gpu_parity import boundary_data
from vasp_amd.capi import HipBackend

case = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tempfile.mkdtemp(), T="0.003")
ns, desc, bc_values, pressure, hook = case
mesh = ns["mesh"]; N2 = mesh.num_nodes
gold = np.load(GOLDEN / "cylinder_tight.npz")["states"]
for mg in sys.argv[1:] or ["1", "0"]:
    os.environ["FSI_DD_MG"] = mg
    hb = HipBackend(desc, lin_rtol=1e-11)
    for k in range(3):
        g, P = boundary_data(case, 1e-3 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=1e-11, rtol=1e-14, max_it=30, lmbda=1.0, recompute=20, recompute_tstep=20)
        hb.shift()
        U = hb.get_state("n")
        errs = {name: np.linalg.norm(U[sl] - gold[k][sl]) / np.linalg.norm(gold[k][sl])
                for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None)))}
        print("mg", mg, "step", k, {n: f"{e:.2e}" for n, e in errs.items()}, "newton", [(f"{h[0]:.1e}", f"{h[1]:.1e}", h[3], f"{h[4]:.1e}") for h in hist], flush=True)
    hb.close()
