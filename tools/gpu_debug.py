"""Ad-hoc GPU debugging script (run on the GPU box): HIP path vs oracle on the cylinder fixture."""
import sys, time, io, contextlib
sys.path.insert(0, ".")
import numpy as np
from vasp_amd.monolithic import prepare
from vasp_amd.capi import HipBackend
from oracle.fsi_oracle import FsiOracle

case = sys.argv[1] if len(sys.argv) > 1 else "cylinder"
mesh_path = {"cylinder": "tests/golden/cylinder/cylinder.h5", "offset_stenosis": "tests/golden/offset_stenosis/offset_stenosis.h5"}[case]
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, bc_values, pressure, hook = prepare(["-p", case, "-dt", "0.001", "-T", "0.002", "--theta", "0.51", "--verbose", "False",
                                                   "--folder", "/tmp/dbg", "--sub-folder", "1", "--new-arguments", f"mesh_path={mesh_path}"])
mesh = ns["mesh"]
t0 = time.time(); hb = HipBackend(desc); print("create %.2fs ndof %d nnz %d" % (time.time() - t0, hb.ndof, hb.lib.fsi_matrix_nnz(hb.ctx)))
o = FsiOracle(desc)
rng = np.random.default_rng(0)
h = mesh.hmin()
U = np.zeros(o.ndof); U1 = np.zeros(o.ndof)
N2 = mesh.num_nodes
U[:3 * N2] = 0.02 * h * rng.standard_normal(3 * N2); U1[:3 * N2] = U[:3 * N2] + 0.002 * h * rng.standard_normal(3 * N2)
U[3 * N2:6 * N2] = 0.1 * rng.standard_normal(3 * N2); U1[3 * N2:6 * N2] = U[3 * N2:6 * N2] + 0.01 * rng.standard_normal(3 * N2)
U[6 * N2:] = 10 * rng.standard_normal(mesh.num_vertices)
with contextlib.redirect_stdout(io.StringIO()):
    ns["t"] = 0.05; hook("pre_solve")(**ns)
g = bc_values(); P = float(pressure.P)
hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
nrm = hb.assemble_residual()
b_gpu = hb.get_state("b")
b_ref = o.rhs(U, U1, P, g)
print("residual: |b| gpu %.15e ref %.15e  max rel diff %.3e" % (nrm, np.linalg.norm(b_ref), np.abs(b_gpu - b_ref).max() / np.abs(b_ref).max()))
for name, sl in [("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))]:
    print("   block", name, "rel diff %.3e" % (np.abs(b_gpu[sl] - b_ref[sl]).max() / (np.abs(b_ref[sl]).max() + 1e-300)))
# Jacobian
o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
t0 = time.time(); A_ref = o.jacobian(U, U1); print("oracle jacobian %.1fs" % (time.time() - t0))
t0 = time.time(); hb.assemble_jacobian(); print("gpu jacobian+ilu %.3fs" % (time.time() - t0)); A_gpu = hb.matrix()
D = (A_gpu - A_ref).tocsr()
rowmax = np.maximum(np.abs(A_ref).max(axis=1).toarray().ravel(), 1e-300)
rel = np.abs(D).max(axis=1).toarray().ravel() / rowmax
print("jacobian: max row-relative diff %.3e  (worst row %d)" % (rel.max(), rel.argmax()))
x = rng.standard_normal(o.ndof); y_gpu = hb.spmv(x); y_ref = A_ref @ x
print("spmv rel diff %.3e" % (np.abs(y_gpu - y_ref).max() / np.abs(y_ref).max()))
print("timers", hb.timers())
# linear solve
import scipy.sparse.linalg as spla
for solver in (0, 1):
    try:
        t0 = time.time(); it, rr = hb.solve(lin_rtol=1e-10, lin_max_it=3000, lin_solver=solver); dt = time.time() - t0
        du = hb.get_state("du"); du_ref = spla.splu(A_ref.tocsc()).solve(b_ref)
        print("solver %d: its %d relres %.2e time %.2fs  |du-du_ref|/|du_ref| = %.3e" % (solver, it, rr, dt, np.linalg.norm(du - du_ref) / np.linalg.norm(du_ref)))
        if solver == 0:
            t0 = time.time(); it, rr = hb.solve(lin_rtol=1e-10, lin_max_it=3000, lin_solver=0); print("   re-solve with recycled space: its %d relres %.2e time %.3fs" % (it, rr, time.time() - t0))
    except Exception as e:
        print("solver", solver, "failed:", e)
print("timers", hb.timers())
