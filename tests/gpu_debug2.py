import sys, time, io, contextlib
sys.path.insert(0, ".")
import numpy as np
print("start", flush=True)
from vasp_amd.monolithic import prepare
from vasp_amd.capi import HipBackend
case = "cylinder"
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, bc_values, pressure, hook = prepare(["-p", case, "-dt", "0.001", "-T", "0.002", "--theta", "0.51", "--verbose", "False",
                                                   "--folder", "/tmp/dbg", "--sub-folder", "1", "--new-arguments", "mesh_path=tests/golden/cylinder/cylinder.h5"])
print("prepared", flush=True)
t0 = time.time(); hb = HipBackend(desc); print("create %.2fs ndof %d" % (time.time() - t0, hb.ndof), flush=True)
with contextlib.redirect_stdout(io.StringIO()):
    ns["t"] = 0.05; hook("pre_solve")(**ns)
hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
t0 = time.time(); nrm = hb.assemble_residual(); print("residual %.3fs norm %.6e" % (time.time() - t0, nrm), flush=True)
t0 = time.time(); hb.assemble_jacobian(); print("jacobian+ilu %.3fs" % (time.time() - t0), flush=True)
print(hb.timers(), flush=True)
t0 = time.time(); it, rr = hb.solve(lin_rtol=1e-8, lin_max_it=50, lin_solver=0) if False else (0, 0)
try:
    t0 = time.time(); it, rr = hb.solve(lin_rtol=1e-8, lin_max_it=20, lin_solver=0); print("solve its %d rr %.3e %.3fs" % (it, rr, time.time() - t0), flush=True)
except Exception as e:
    print("solve:", e, "%.3fs" % (time.time() - t0), flush=True)
print(hb.timers(), flush=True)
