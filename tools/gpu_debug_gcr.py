"""Replays the known-answer test's five steps on the stenosis fixture and prints the Newton / Krylov history."""
import contextlib, io, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import prepare_case, GOLDEN
from vasp_amd.capi import HipBackend, FsiError
case = prepare_case("offset_stenosis", GOLDEN / "offset_stenosis" / "offset_stenosis.h5", "/tmp/dbg_os", dt="0.01", T="0.04")
ns, desc, bc_values, pressure, hook = case
hb = HipBackend(desc, lin_rtol=1e-9)
for k in range(5):
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 0.01 * (k + 1); hook("pre_solve")(**ns)
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
    try:
        h = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                            recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
    except FsiError as e:
        print("step", k, "FAILED", e); print(hb.history[-1] if hb.history else None); break
    print("step", k, [(f"{a:.2e}", f"{b:.2e}", c, d, f"{e:.1e}") for a, b, c, d, e in h], flush=True)
    hb.shift()
print(hb.timers())
