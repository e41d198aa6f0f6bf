"""GPU box helper: first linear solve of a case (zero state, first-step boundary data) with several inner-solve settings."""
import sys, time, io, contextlib, json
sys.path.insert(0, ".")
import numpy as np
from vasp_amd.monolithic import prepare
from vasp_amd.capi import HipBackend

problem, mesh_path, dt = sys.argv[1], sys.argv[2], sys.argv[3]
settings = [tuple(float(x) for x in s.split(",")) for s in sys.argv[4:]] or [(0, 1e-2, 40)]
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, bc_values, pressure, hook = prepare(["-p", problem, "-dt", dt, "-T", "1", "--theta", "0.51", "--verbose", "False",
                                                   "--folder", "/tmp/lin", "--sub-folder", "1", "--new-arguments", f"mesh_path={mesh_path}"])
    ns["t"] = float(dt); hook("pre_solve")(**ns)
t0 = time.time(); hb = HipBackend(desc); print("create %.2fs ndof %d nnz %d" % (time.time() - t0, hb.ndof, hb.lib.fsi_matrix_nnz(hb.ctx)), flush=True)
hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
for precond, rtol, maxit in settings:
    hb.set_linear_solver(int(precond), rtol, int(maxit))
    nrm = hb.assemble_residual()
    t0 = time.time(); hb.assemble_jacobian(); tj = time.time() - t0
    hb.timers(reset=True)
    t0 = time.time()
    try:
        it, rr = hb.solve(lin_rtol=1e-8, lin_max_it=int(sys.argv[0] and 150), lin_solver=0); msg = "ok"
    except Exception as e:
        it, rr, msg = -1, -1, str(e)[:90]
    tm = hb.timers()
    print("precond %d inner rtol %.0e maxit %d | jac+fac %.2fs | outer its %s relres %.2e  %.2fs | inner vv/S/dd per apply: %.1f %.1f %.1f | prec %.0f ms spmv %.0f ms ortho %.0f ms | %s"
          % (precond, rtol, maxit, tj, it, rr, time.time() - t0, tm["inner_vv_iters"] / max(1, tm["precond_applies"]),
             tm["inner_schur_iters"] / max(1, tm["precond_applies"]), tm["inner_dd_iters"] / max(1, tm["precond_applies"]),
             tm["precond_ms"], tm["spmv_ms"], tm["ortho_ms"], msg), flush=True)
