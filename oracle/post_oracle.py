"""CPU ORACLE (test infrastructure, not product code) for the solid stress / strain and wall-shear-stress kernels
(SURVEY.md §8f row f4).

Restates, in numpy, what VaSP's FEniCS post-processing computes per saved time step:

* ``compute_stress_strain`` [REF src/vasp/postprocessing/postprocessing_fenics/compute_stress_strain.py:188-263]:
  Cauchy stress ``1/J F S F^T`` and Green-Lagrange strain ``E`` of the P2 displacement, L2-projected cell by cell onto
  the tensor DG1 space over the solid cells (``solve_dg``), then the largest principal value of each projected tensor
  (``common.get_eig``) projected onto scalar DG1 (``project_dg`` [REF postprocessing_fenics_common.py:32-54]).
  ``common.F_, E, S, J_`` are the constitutive routines of the solid residual (oracle/fsi_oracle.py); ``get_eig`` lives in
  the un-vendored turtleFSI (closed-form roots of the characteristic polynomial after Kopp / Cardano); mathematically it
  is the largest eigenvalue of a symmetric 3x3 tensor, which is what ``numpy.linalg.eigvalsh`` returns here.
* ``Stress`` of compute_hemodynamics [REF src/vasp/postprocessing/postprocessing_fenics/compute_hemodynamics.py:120-157]:
  ``sigma = 2 mu sym(grad u)``, ``F = -sigma n``, tangential part ``Ft = F - (F.n) n`` on the exterior facets of the
  fluid mesh, projected with the surface mass matrix onto the DG1 space of the boundary cells (``SurfaceProjector``
  :91-117: ``inner(u, v) * ds``, zero rows -> identity), read out at the three vertices of every boundary facet
  (``InterpolateDG``).

Quadrature: the degree-6 rules used everywhere else (the integrands are rational in grad d; the reference leaves the
degree to FFC's estimate there, so the last digits of a FEniCS run are not pinned by anything in the tree).
Parity status: no reference test pins a number for these outputs that is reachable without the missing large blobs
(REF tests/test_compute_hemodynamics.py needs u.h5 from .MISSING_LARGE_BLOBS): parity unpinned; the HIP kernels are held
against this restatement, and this restatement against closed-form cases in tests/test_post_oracle.py.
"""
from __future__ import annotations

import numpy as np

from .fsi_oracle import I3, _det3, _inv3, keast24, tabulate_p2, triangle12


def second_piola(g, mu, lam, model=0, C10=0.0, C01=0.0, C11=0.0):
    """S(d) for grad d = g (..., 3, 3): StVenantKirchoff or the compressible Mooney-Rivlin energy of fsi_oracle.py."""
    F = I3 + g
    C = np.swapaxes(F, -1, -2) @ F
    if model == 1:
        Ci, _ = _inv3(C)
        J = _det3(F)
        I1 = np.trace(C, axis1=-2, axis2=-1)
        I2 = 0.5 * (I1 ** 2 - np.einsum("...ij,...ji->...", C, C))
        Jm23 = J ** (-2.0 / 3.0)
        I1b, I2b = Jm23 * I1, Jm23 ** 2 * I2
        a1 = (2.0 * (C10 + C11 * (I2b - 3.0)) * Jm23)[..., None, None]
        a2 = (2.0 * (C01 + C11 * (I1b - 3.0)) * Jm23 ** 2)[..., None, None]
        K = lam + 2.0 * mu / 3.0
        return (a1 * (I3 - (I1 / 3.0)[..., None, None] * Ci)
                + a2 * (I1[..., None, None] * I3 - C - (2.0 / 3.0) * I2[..., None, None] * Ci)
                + (K * np.log(J) * J)[..., None, None] * Ci)
    E = 0.5 * (C - I3)
    return lam * np.trace(E, axis1=-2, axis2=-1)[..., None, None] * I3 + 2.0 * mu * E


def stress_strain_dg1(coords, tets, tet_nodes, d_nodal, cells, props, model=0, eig="eigvalsh"):
    """DG1 coefficients on ``cells`` (solid cells of one region; props = (rho, mu, lambda[, C10, C01, C11])).

    Returns dict: TrueStress (n,4,3,3), GreenLagrangeStrain (n,4,3,3), MaxPrincipalStress (n,4), MaxPrincipalStrain (n,4);
    coefficient a belongs to local vertex a of the cell (DG1 = P1 on the cell, nodal basis).
    ``eig``: "eigvalsh" (LAPACK) or "kopp" (the trigonometric closed form of turtleFSI's ``get_eig``, which carries ~1e-8 of
    the tensor's magnitude in round-off when two principal values are close - the formula's conditioning, not a bug)."""
    qp, qw = keast24()
    N, dNref, L, dL = tabulate_p2(qp)
    xc = coords[tets[cells]]
    Jm = np.stack([xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0], xc[:, 3] - xc[:, 0]], axis=2)
    Jinv, det = _inv3(Jm)
    w = np.abs(det)[:, None] * qw[None, :]
    G = np.einsum("qak,ckj->cqaj", dNref, Jinv)
    g = np.einsum("cai,cqaj->cqij", d_nodal[tet_nodes[cells]], G)              # grad d at the quadrature points
    F = I3 + g
    J = _det3(F)
    mu, lam = props[1], props[2]
    extra = dict(C10=props[3], C01=props[4], C11=props[5]) if len(props) >= 6 else {}
    S = second_piola(g, mu, lam, model, **extra)
    sigma = (F @ S @ np.swapaxes(F, -1, -2)) / J[..., None, None]
    E = 0.5 * (np.swapaxes(F, -1, -2) @ F - I3)
    M = np.einsum("cq,qa,qb->cab", w, L, L)                                     # P1 mass matrix of the cell

    def project(f):                                                             # f (n,Q,...) -> (n,4,...)
        rhs = np.einsum("cq,qa,cq...->ca...", w, L, f)
        flat = rhs.reshape(rhs.shape[0], 4, -1)
        return np.linalg.solve(M, flat).reshape(rhs.shape)

    TS, GLS = project(sigma), project(E)

    def max_principal(Tc):                                                      # DG1 tensor -> DG1 largest eigenvalue
        Tq = np.einsum("qa,caij->cqij", L, Tc)
        Tq = 0.5 * (Tq + np.swapaxes(Tq, -1, -2))
        return project(kopp_max_eigenvalue(Tq) if eig == "kopp" else np.linalg.eigvalsh(Tq)[..., -1])

    return dict(TrueStress=TS, GreenLagrangeStrain=GLS, MaxPrincipalStress=max_principal(TS),
                MaxPrincipalStrain=max_principal(GLS))


def kopp_max_eigenvalue(T):
    """Largest root of the characteristic polynomial of a symmetric 3x3 tensor by the trigonometric (Cardano) form used
    by turtleFSI's ``get_eig`` (J. Kopp, eqs. 21-34), with that routine's perturbations of p, q and the discriminant."""
    tol1, tol2, tol3 = 1e-16, 1e-24, 1e-40
    I1 = np.trace(T, axis1=-2, axis2=-1)
    I2 = 0.5 * (I1 ** 2 - np.einsum("...ij,...ij->...", T, T))
    I3_ = _det3(T)
    p = I1 ** 2 - 3 * I2
    p = np.where(p < tol1, np.abs(p) + 2 * tol1, p)
    q = 27 / 2 * I3_ + I1 ** 3 - 9 / 2 * I1 * I2
    q = np.where(np.abs(q) < tol2, q + np.sign(q) * 2 * tol2, q)
    nom2 = 27 * (1 / 4 * I2 ** 2 * (p - I2) + I3_ * (27 / 4 * I3_ - q))
    nom2 = np.where(nom2 < tol3, np.abs(nom2) + 2 * tol3, nom2)
    phi = 1 / 3 * np.arctan2(np.sqrt(nom2), q)
    return 1 / 3 * (np.sqrt(p) * 2 * np.cos(phi) + I1)


def wall_shear_stress(coords, tets, tet_nodes, v_nodal, facet_cells, facet_local, mu):
    """Tangential traction on exterior facets, DG1-projected per boundary cell.

    ``facet_cells`` (nf,), ``facet_local`` (nf,): the cell of each exterior facet and the local index of the vertex
    opposite to it (UFC facet numbering).  Returns (nf, 3, 3): the projected ``Ft`` at the three facet vertices in the
    order of the cell's local vertices with ``facet_local`` removed."""
    facet_cells = np.asarray(facet_cells)
    facet_local = np.asarray(facet_local)
    tp, tw = triangle12()
    lt = np.stack([1 - tp[:, 0] - tp[:, 1], tp[:, 0], tp[:, 1]], axis=1)         # (12,3) facet barycentric
    ucells, inv = np.unique(facet_cells, return_inverse=True)
    nc = len(ucells)
    xc = coords[tets[ucells]]
    Jm = np.stack([xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0], xc[:, 3] - xc[:, 0]], axis=2)
    Jinv, _ = _inv3(Jm)
    Mc = np.zeros((nc, 4, 4))
    bc = np.zeros((nc, 4, 3))
    verts_of = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
    for f in range(len(facet_cells)):
        c, lf = inv[f], facet_local[f]
        lv = verts_of[lf]
        x = xc[c][lv]
        nv = np.cross(x[1] - x[0], x[2] - x[0])
        area = 0.5 * np.linalg.norm(nv)
        n = nv / (2 * area)
        if np.dot(n, x[0] - xc[c][lf]) < 0:
            n = -n                                                               # outward: away from the opposite vertex
        lam = np.zeros((12, 4))
        lam[:, lv] = lt
        _, dNref, _, _ = tabulate_p2(lam[:, 1:4])
        G = np.einsum("qak,kj->qaj", dNref, Jinv[c])
        gv = np.einsum("ai,qaj->qij", v_nodal[tet_nodes[ucells[c]]], G)
        sig = mu * (gv + np.swapaxes(gv, -1, -2))                               # 2 mu sym(grad u)
        Fv = -np.einsum("qij,j->qi", sig, n)
        Ft = Fv - (Fv @ n)[:, None] * n[None, :]
        wq = 2.0 * area * tw
        Mc[c] += np.einsum("q,qa,qb->ab", wq, lam, lam)
        bc[c] += np.einsum("q,qa,qi->ai", wq, lam, Ft)
    zero = np.abs(Mc).sum(axis=2) == 0.0                                         # ident_zeros on the surface mass matrix
    for c in range(nc):
        for a in np.nonzero(zero[c])[0]:
            Mc[c, a, a] = 1.0
    Xc = np.linalg.solve(Mc, bc)
    out = np.empty((len(facet_cells), 3, 3))
    for f in range(len(facet_cells)):
        out[f] = Xc[inv[f]][verts_of[facet_local[f]]]
    return out
