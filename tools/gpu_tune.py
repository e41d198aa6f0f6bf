"""GPU box helper: first linear solve of a generated case for several Chebyshev settings 'itsS,kappaS,itsF,kappaF,itsP,kappaP'."""
import sys, time, io, contextlib
sys.path.insert(0, ".")
import numpy as np
from vasp_amd.monolithic import prepare
from vasp_amd.capi import HipBackend
problem, mesh_path, dt = sys.argv[1], sys.argv[2], sys.argv[3]
settings = [tuple(float(x) for x in s.split(",")) for s in sys.argv[4:]]
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, bc_values, pressure, hook = prepare(["-p", problem, "-dt", dt, "-T", "1", "--theta", "0.501", "--verbose", "False",
                                                   "--folder", "/tmp/lin", "--sub-folder", "1", "--new-arguments", f"mesh_path={mesh_path}"])
    ns["t"] = float(dt); hook("pre_solve")(**ns)
t0 = time.time(); hb = HipBackend(desc); print("create %.2fs ndof %d nnz %d" % (time.time() - t0, hb.ndof, hb.lib.fsi_matrix_nnz(hb.ctx)), flush=True)
hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
hb.assemble_residual(); hb.assemble_jacobian()
for st in settings:
    hb.set_chebyshev(*st)
    hb.timers(reset=True)
    t0 = time.time()
    try:
        it, rr = hb.solve(lin_rtol=1e-8, lin_max_it=400, lin_solver=0); msg = "ok"
    except Exception as e:
        it, rr, msg = -1, -1, str(e)[:90]
    tm = hb.timers()
    print("cheb %s | outer its %s relres %.2e  %.2fs | prec %.0f ms (%.1f ms/apply) spmv %.0f ms ortho %.0f ms dd its/apply %.1f | %s"
          % (st, it, rr, time.time() - t0, tm["precond_ms"], tm["precond_ms"] / max(1, tm["precond_applies"]), tm["spmv_ms"], tm["ortho_ms"],
             tm["inner_dd_iters"] / max(1, tm["precond_applies"]), msg), flush=True)
