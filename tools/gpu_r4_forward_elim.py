"""Round 4: the block factorisation of the preconditioner eliminates the solid displacement with the nodal relation d = k theta v
in the OPERATOR (A_vv~ = A_vv + k theta A_vd on solid columns) but not in the RIGHT-HAND SIDE: with a residual r_d on the solid rows
the exact elimination gives r_v' = r_v - A_vd A_dd^-1 r_d (and the same for the pressure rows).  The first Newton iteration of every
time step has its residual mostly in r_d.  Prototype on the host: GCR (right preconditioning, scaled rows as in the library) on the
first linear system of a time step with M^-1 as it is and with the forward elimination in front of it (g = A_dd^-1 r_d on the solid
nodes: exact, diagonal, a few Jacobi sweeps).

    python tools/gpu_r4_forward_elim.py TETS
"""
import contextlib, io, sys, tempfile
from pathlib import Path
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def gcr(A, b, prec, tol, maxit):
    x = np.zeros_like(b); r = b.copy(); bn = np.linalg.norm(b)
    Q, P, hist = [], [], [1.0]
    for it in range(maxit):
        z = prec(r)
        w = A @ z
        for _ in range(2):
            for q, p in zip(Q, P):
                h = q @ w
                w -= h * q; z -= h * p
        wn = np.linalg.norm(w)
        q, p = w / wn, z / wn
        a = q @ r
        x += a * p; r -= a * q
        Q.append(q); P.append(p)
        hist.append(np.linalg.norm(r) / bn)
        if hist[-1] <= tol:
            break
    return x, hist


def main():
    tets = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
    from vasp_amd.capi import HipBackend
    from vasp_amd.meshgen import write_mesh
    from vasp_amd.monolithic import advance, prepare
    tmp = Path(tempfile.mkdtemp())
    mp = tmp / "stenosis.h5"
    write_mesh(mp, tets, seed=0)
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(["-p", "offset_stenosis", "-dt", "0.001", "-T", "0.01", "--theta", "0.501", "--verbose", "False",
                                                       "--folder", str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={mp}"])
    hb = HipBackend(desc)
    for which, fn in ns["dvp_"].items():
        fn.backend, fn.which = hb, which
    ns["backend"] = hb
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            advance(ns, hb, bc_values, pressure, hook, 0, out=lambda *a: None)
            ns["counter"] += 1
        ns["t"] = ns["t"] + float(ns["dt"])
        ns.update(hook("pre_solve")(**ns) or {})
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
    mesh = ns["mesh"]
    N2 = mesh.num_nodes
    n = hb.ndof
    kind = np.asarray(desc["cell_kind"]); tn = np.asarray(mesh.tet_nodes)
    solid = np.zeros(N2, bool); solid[np.unique(tn[kind == 1])] = True
    for label in ("first Newton iteration of the step", "second Newton iteration (after one exact update)"):
        hb.assemble_jacobian()
        hb.assemble_residual()
        b = hb.get_state("b")
        A = hb.matrix().tocsr()
        D = 1.0 / np.maximum(abs(A).max(axis=1).toarray().ravel(), 1e-300)
        As = sp.diags(D) @ A
        bs = D * b
        sd = (3 * np.flatnonzero(solid)[:, None] + np.arange(3)[None, :]).ravel()          # d dofs of the solid nodes
        vrows = np.arange(3 * N2, 6 * N2); prows = np.arange(6 * N2, n)
        Add = A[sd][:, sd].tocsc()
        Avd = A[vrows][:, sd].tocsr(); Apd = A[prows][:, sd].tocsr()
        lu = spla.splu(Add)
        dg = Add.diagonal()
        print(f"== {label}: |b| per field (scaled rows): d {np.linalg.norm(bs[:3 * N2]):.3e}  v {np.linalg.norm(bs[3 * N2:6 * N2]):.3e}  p {np.linalg.norm(bs[6 * N2:]):.3e};"
              f" solid d dofs {len(sd)}, nnz(A_vd) {Avd.nnz}, nnz(A_pd) {Apd.nnz}", flush=True)

        def plain(rs):
            return hb.apply_preconditioner(rs / D)

        def make(gsolve):
            def f(rs):
                ru = rs / D
                g = gsolve(ru[sd])
                ru = ru.copy()
                ru[vrows] -= Avd @ g
                ru[prows] -= Apd @ g
                return hb.apply_preconditioner(ru)
            return f

        def jacobi(k):
            def f(r):
                g = np.zeros_like(r)
                for _ in range(k):
                    g += 0.8 * (r - Add @ g) / dg
                return g
            return f

        for name, prec in (("M^-1 as it is", plain), ("forward elimination, exact A_dd^-1", make(lu.solve)), ("forward elimination, diagonal", make(lambda r: r / dg)),
                           ("forward elimination, 3 Jacobi sweeps", make(jacobi(3))), ("forward elimination, 6 Jacobi sweeps", make(jacobi(6)))):
            x, hist = gcr(As, bs, prec, 1e-8, 80)
            first = lambda t: next((i for i, h in enumerate(hist) if h <= t), None)
            print(f"  {name:40s} iterations to 1e-2 / 1e-4 / 1e-6 / 1e-8: {first(1e-2)} / {first(1e-4)} / {first(1e-6)} / {first(1e-8)}   first five: " + " ".join(f"{h:.2e}" for h in hist[1:6]), flush=True)
        # move on: one (nearly exact) Newton update, then the next system of the same step
        x, _ = gcr(As, bs, plain, 1e-10, 120)
        U = hb.get_state("n")
        hb.set_state("n", U + x)
    hb.close()


if __name__ == "__main__":
    main()
