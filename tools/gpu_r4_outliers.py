"""Round 4: the first solve after every Jacobian refresh spends 15 - 20 iterations without progress (|w'| / |w| ~ 1e-5: A M^-1 r lies
in the span of a handful of directions).  Which modes of A M^-1 are these?  Block power iteration (Rayleigh-Ritz on a growing Krylov
space of A M^-1, via fsi_apply_preconditioner + fsi_spmv) on the bench workload after one time step, and the share of the leading
vectors per field (d / v / p) and per region (fluid / solid / interface).

    python tools/gpu_r4_outliers.py TETS > gpurun_out/r04_outliers.txt
"""
import contextlib, io, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


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
        for _ in range(2):
            advance(ns, hb, bc_values, pressure, hook, 0, out=lambda *a: None)
            ns["counter"] += 1
    mesh = ns["mesh"]
    N2, V = mesh.num_nodes, mesh.num_vertices
    n = hb.ndof
    kind = np.asarray(desc["cell_kind"])
    tn = np.asarray(mesh.tet_nodes)
    in_solid = np.zeros(N2, bool); in_solid[np.unique(tn[kind == 1])] = True
    in_fluid = np.zeros(N2, bool); in_fluid[np.unique(tn[kind == 0])] = True
    iface = in_solid & in_fluid
    bc = np.zeros(n, bool); bc[np.asarray(desc["bc_dofs"])] = True
    print(f"{len(tn)} tets, {n} dofs, {iface.sum()} interface nodes, {bc.sum()} Dirichlet dofs", flush=True)

    def op(x):
        return hb.spmv(hb.apply_preconditioner(x))

    # --- the actual right-hand side of the next time step's first Newton iteration (a fresh Jacobian first, as after a refresh) ---
    def shares(vec):
        d, vv, p = vec[:3 * N2].reshape(N2, 3), vec[3 * N2:6 * N2].reshape(N2, 3), vec[6 * N2:]
        return {"d solid": np.linalg.norm(d[in_solid & ~iface]), "d iface": np.linalg.norm(d[iface]), "d fluid": np.linalg.norm(d[in_fluid & ~iface]),
                "v solid": np.linalg.norm(vv[in_solid & ~iface]), "v iface": np.linalg.norm(vv[iface]), "v fluid": np.linalg.norm(vv[in_fluid & ~iface]),
                "p": np.linalg.norm(p)}
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = ns["t"] + float(ns["dt"])
        ns.update(hook("pre_solve")(**ns) or {})
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
    hb.assemble_jacobian()
    hb.assemble_residual()
    bvec = hb.get_state("b")
    D = hb.spmv  # (unscaled product)
    z = hb.apply_preconditioner(bvec)
    e = hb.spmv(z) - bvec
    print("the Newton right-hand side b, z = M^-1 b and e = A z - b, 2-norms per block (user units, rows unscaled):")
    for tag, vec in (("b", bvec), ("z", z), ("A z", e + bvec), ("e", e)):
        print(f"    {tag:4s} " + "  ".join(f"{k} {v_:.3e}" for k, v_ in shares(vec).items()), flush=True)
    # the same after one exact-ish correction: how far is M^-1 from A^-1 on b, field by field?  cos of the angle between A z and b
    Az = e + bvec
    print(f"    cos(A z, b) = {Az @ bvec / np.linalg.norm(Az) / np.linalg.norm(bvec):.3e}   |A z| / |b| = {np.linalg.norm(Az) / np.linalg.norm(bvec):.3e}")
    # block by block: feed b restricted to one block
    for name, mask in (("d rows", np.r_[np.ones(3 * N2, bool), np.zeros(n - 3 * N2, bool)]), ("v rows", np.r_[np.zeros(3 * N2, bool), np.ones(3 * N2, bool), np.zeros(n - 6 * N2, bool)]),
                       ("p rows", np.r_[np.zeros(6 * N2, bool), np.ones(n - 6 * N2, bool)])):
        bb = np.where(mask, bvec, 0.0)
        if not np.any(bb):
            continue
        zz = hb.apply_preconditioner(bb)
        ee = hb.spmv(zz) - bb
        print(f"  b restricted to the {name}: |b| {np.linalg.norm(bb):.3e}")
        for tag, vec in (("z", zz), ("e", ee)):
            print(f"    {tag:4s} " + "  ".join(f"{k} {v_:.3e}" for k, v_ in shares(vec).items()), flush=True)

    rng = np.random.default_rng(0)
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    Q = np.zeros((n, 0))
    v = rng.standard_normal(n); v[bc] = 0.0
    Hs = []
    # Arnoldi on A M^-1
    H = np.zeros((K + 1, K))
    Qs = [v / np.linalg.norm(v)]
    for k in range(K):
        w = op(Qs[k])
        for j in range(k + 1):
            H[j, k] = Qs[j] @ w
            w -= H[j, k] * Qs[j]
        for j in range(k + 1):                     # second pass
            c = Qs[j] @ w
            H[j, k] += c
            w -= c * Qs[j]
        H[k + 1, k] = np.linalg.norm(w)
        Qs.append(w / H[k + 1, k])
        if (k + 1) % 10 == 0:
            ev = np.linalg.eigvals(H[:k + 1, :k + 1])
            ev = ev[np.argsort(-np.abs(ev))]
            print(f"Arnoldi step {k + 1}: leading Ritz values of A M^-1:", " ".join(f"{e.real:.3g}{'' if abs(e.imag) < 1e-9 * abs(e) else f'+{e.imag:.2g}i'}" for e in ev[:12]), "... smallest", " ".join(f"{abs(e):.3g}" for e in ev[-4:]), flush=True)
    ev, S = np.linalg.eig(H[:K, :K])
    order = np.argsort(-np.abs(ev))
    Qm = np.stack(Qs[:K], axis=1)
    names = ("d", "v", "p")
    for idx in list(order[:4]) + list(order[-10:]):
        y = (Qm @ S[:, idx]).real
        y /= np.linalg.norm(y)
        # the mode as a right-hand side; and what the preconditioner makes of it
        z = hb.apply_preconditioner(y)
        parts = {}
        for vec, tag in ((y, "r"), (z, "M^-1 r")):
            d, vv, p = vec[:3 * N2].reshape(N2, 3), vec[3 * N2:6 * N2].reshape(N2, 3), vec[6 * N2:]
            tot = np.linalg.norm(vec) ** 2
            parts[tag] = {"d solid": (d[in_solid & ~iface] ** 2).sum() / tot, "d iface": (d[iface] ** 2).sum() / tot, "d fluid": (d[in_fluid & ~iface] ** 2).sum() / tot,
                          "v solid": (vv[in_solid & ~iface] ** 2).sum() / tot, "v iface": (vv[iface] ** 2).sum() / tot, "v fluid": (vv[in_fluid & ~iface] ** 2).sum() / tot,
                          "p": (p ** 2).sum() / tot}
        print(f"Ritz value {ev[idx].real:10.4g}: |M^-1 r| / |r| = {np.linalg.norm(z):.3g}")
        for tag, pr in parts.items():
            print(f"    {tag:7s} " + "  ".join(f"{k} {v_:.3f}" for k, v_ in pr.items()))
    hb.close()


if __name__ == "__main__":
    main()
