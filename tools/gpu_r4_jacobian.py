"""Round 4, N1 in place: the refresh kernel of the Jacobian (F_nonlinear, rows v and p) at bench size with the element
contraction on the vector pipe (k_jacobian<2, W>) and on the matrix pipe (k_jacobian_mfma<W>), W = waves per SIMD the
register budget is set for.  Each variant: the assembled matrices must agree to round-off, then the kernel time of three
refreshes (FsiTimers.jacobian_ms: HIP events around the 59 colour launches + matrix finish).

    python tools/gpu_r4_jacobian.py [tets] > gpurun_out/r04_jacobian_mfma.txt
"""
import contextlib
import io
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    tets = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    from vasp_amd.capi import HipBackend
    from vasp_amd.meshgen import write_mesh
    from vasp_amd.monolithic import prepare
    tmp = Path(tempfile.mkdtemp())
    write_mesh(tmp / "m.h5", tets, seed=0)
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(
            ["-p", "offset_stenosis", "-dt", "0.001", "-T", "0.01", "--theta", "0.501", "--verbose", "False", "--folder", str(tmp / "r"),
             "--sub-folder", "1", "--new-arguments", f"mesh_path={tmp / 'm.h5'}"])
        ns["t"] = 0.005
        hook("pre_solve")(**ns)
    os.environ["FSI_KRYLOV_CAP"] = "8"                       # no Krylov store needed here
    mesh = ns["mesh"]
    rng = np.random.default_rng(0)
    N2, h = mesh.num_nodes, mesh.hmin()
    U = np.zeros(mesh.num_dofs)
    U[:3 * N2] = 0.01 * h * rng.standard_normal(3 * N2)
    U[3 * N2:6 * N2] = 0.1 * rng.standard_normal(3 * N2)
    U[6 * N2:] = 10 * rng.standard_normal(mesh.num_vertices)
    x = rng.standard_normal(mesh.num_dofs)
    ref = None
    print(f"{mesh.num_cells} tets, {mesh.num_dofs} dofs")
    for name, env in (("k_jacobian<2,2> (vector pipe, shipped)", {"FSI_JAC_MFMA": "0", "FSI_JAC_WAVES": "2"}),
                      ("k_jacobian<2,1> (vector pipe, 1 wave/SIMD)", {"FSI_JAC_MFMA": "0", "FSI_JAC_WAVES": "1"}),
                      ("k_jacobian_mfma<2> (matrix pipe)", {"FSI_JAC_MFMA": "1", "FSI_JAC_WAVES": "2"}),
                      ("k_jacobian_mfma<1> (matrix pipe, 1 wave/SIMD)", {"FSI_JAC_MFMA": "1", "FSI_JAC_WAVES": "1"})):
        os.environ.update(env)                               # FsiTuning.jacobian_waves / jacobian_mfma, read by fsi_create
        hb = HipBackend(desc)
        hb.set_state("n", U); hb.set_state("n-1", 0.9 * U)
        hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
        hb.assemble_jacobian()                               # warm-up
        y = hb.spmv(x)
        if ref is None:
            ref = y
        err = float(np.abs(y - ref).max() / np.abs(ref).max())
        hb.timers(reset=True)
        for _ in range(3):
            hb.assemble_jacobian()
        tm = hb.timers()
        print(f"{name:48s} {tm['jacobian_ms'] / tm['jacobian_calls']:8.2f} ms per refresh (kernel + finish), factor {tm['factor_ms'] / max(1, tm['factor_calls']):7.2f} ms, "
              f"A x vs shipped kernel: {err:.2e}", flush=True)
        hb.close()


if __name__ == "__main__":
    main()
