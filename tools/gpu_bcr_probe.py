"""Round 5: the exact coarse solve of the solid cycle (csrc/fsi_bcr.hip) on a generated offset-stenosis mesh of N tets -
planner figures, accuracy of the production kernels against a sparse LU of the same operator (random and smooth right-hand
sides), and the cost of a refresh.    python tools/gpu_bcr_probe.py [tets]"""
import contextlib
import io
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import scipy.sparse.linalg as spla

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from vasp_amd.capi import HipBackend  # noqa: E402
from vasp_amd.meshgen import write_mesh  # noqa: E402
from vasp_amd.monolithic import prepare  # noqa: E402

tets = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
tmp = Path(tempfile.mkdtemp())
write_mesh(tmp / "s.h5", tets, seed=0)
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, bc_values, pressure, hook = prepare(["-p", "offset_stenosis", "-dt", "0.001", "-T", "0.02", "--theta", "0.501", "--verbose", "False",
                                                   "--folder", str(tmp / "r"), "--sub-folder", "1", "--new-arguments", f"mesh_path={tmp / 's.h5'}"])
    ns["t"] = 0.001
    hook("pre_solve")(**ns)
hb = HipBackend(desc)
hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
hb.assemble_residual()
for k in range(2):
    t0 = time.perf_counter()
    hb.assemble_jacobian()
    print(f"assemble_jacobian + refresh #{k}: {1e3 * (time.perf_counter() - t0):.1f} ms")
info = hb.solid_coarse_info()
print(info)
A, cptr, ccol = hb.solid_coarse_matrix()
n = A.shape[0]
lu = spla.splu(A.tocsc())
rng = np.random.default_rng(0)
ev = spla.eigsh((A + A.T) * 0.5, k=3, sigma=0, which="LM", return_eigenvectors=False)
print("smallest eigenvalues of the symmetric part:", ev, " largest diagonal:", A.diagonal().max(), " asymmetry:", abs(A - A.T).max() / abs(A).max())
for name, rhs in (("random", rng.standard_normal(n)), ("A * ones", A @ np.ones(n)), ("lowest modes", None)):
    if rhs is None:
        rhs = lu.solve(rng.standard_normal(n)); rhs /= np.linalg.norm(rhs)      # rich in the lowest modes
    x = hb.solid_coarse_solve(rhs)
    xr = lu.solve(rhs)
    print(f"{name:13s} |A x - b| / |b| = {np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs):.2e}   |x - x_lu| / |x_lu| = {np.linalg.norm(x - xr) / np.linalg.norm(xr):.2e}"
          f"   |x| / |b| = {np.linalg.norm(xr) / np.linalg.norm(rhs):.2e}")
tm = hb.timers()
print("factor_ms", tm["factor_ms"], "calls", tm["factor_calls"])
hb.close()
