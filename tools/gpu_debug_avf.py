"""AVF case on the synthetic tube: which part of the block preconditioner fails its self-test (FSI_DEBUG_PRECOND=1)."""
import contextlib, io, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from conftest import make_avf_case
from vasp_amd.capi import HipBackend, FsiError
case = make_avf_case(Path("/tmp/dbg_avf"))
ns, desc, bc_values, pressure, hook = case
hb = HipBackend(desc, lin_rtol=1e-10)
Z = np.zeros(hb.ndof)
for k in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 1e-4 * (k + 1); hook("pre_solve")(**ns)
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
    try:
        h = hb.newton_solve(counter=k, first_step_num=0, atol=1e-10, rtol=1e-14, max_it=30, lmbda=1.0, recompute=30, recompute_tstep=10)
        print("step", k, [(f"{a:.2e}", f"{b:.2e}", c, d) for a, b, c, d, e in h], flush=True)
    except FsiError as e:
        print("step", k, "FAILED", e, flush=True); break
    hb.shift()
