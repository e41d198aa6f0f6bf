"""CPU ORACLE (test infrastructure, not product code) for the monolithic ALE-FSI Newton step.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the product path (``vasp_amd``) never does.

What it restates
----------------
The hot path of the reference is not in the reference tree: VaSP's problem files hand their
parameters, markers and boundary conditions to the un-vendored solver ``turtleFSI`` (git master,
unpinned; docs floor 2.4.0 [REF environment.yml:25, docs/installation.md:15]) running on legacy
FEniCS/DOLFIN 2019.x + FFC/FIAT + PETSc/MUMPS [REF environment.yml:8].  This file restates, in
numpy, the published algorithm of those packages for exactly the call sites VaSP uses
(Appendix A.2/A.3):

* ``turtleFSI/modules/fluid.py``  ``fluid_setup``      -> ``_fluid_residual``
* ``turtleFSI/modules/solid.py``  ``solid_setup``      -> ``_solid_residual``  (delta = 1e7, StVenantKirchoff)
* ``turtleFSI/modules/laplace.py`` ``extrapolate_setup`` ("constant": alfa = 1) -> in ``_fluid_residual``
* ``turtleFSI/modules/common.py`` ``F_, J_, E, S, Piola1, sigma_f_u, sigma_f_p``
* ``turtleFSI/modules/newtonsolver.py`` ``solver_setup`` / ``newtonsolver`` -> ``solver_setup`` / ``newtonsolver``
* VaSP's own additions: interface pressure on ``dS(fsi_id)``
  [REF src/vasp/simulations/offset_stenosis.py:184-190], Robin term parameters
  [REF src/vasp/simulations/aneurysm.py:73-76]
* DOLFIN semantics: ``DirichletBC.apply`` (rows -> identity, rhs -> g - u), ``ident_zeros``,
  '+' side of an interior facet = cell with the larger domain id, FIAT default degree-6 rules
  (Keast 24-point tetrahedron, 12-point triangle), UFC P2 ordering.

Parity status
-------------
Pinned by the reference's own known-answer test: ``tests/test_simulations.py:22-57`` (offset
stenosis fixture, 5 steps, probe 5 velocity/pressure/displacement, ``np.isclose`` defaults) — see
``tests/test_oracle_pins.py``.

Jacobians are obtained by complex-step differentiation of the element residuals (exact to round-off),
i.e. independently of the hand-derived tangent used by the HIP kernels.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

DELTA = 1.0e7  # turtleFSI/modules/solid.py: penalty that enforces d_t = v in the solid

TET_EDGES = np.array([[2, 3], [1, 3], [1, 2], [0, 3], [0, 2], [0, 1]])


# ------------------------------------------------------------------------------------------------
# FIAT default quadrature for quadrature_degree = 6 [REF offset_stenosis.py:18]
# ------------------------------------------------------------------------------------------------

def keast24():
    """Keast 24-point degree-6 rule on the UFC reference tetrahedron (FIAT `_tetrahedron_scheme(6)`).

    Returns points (24,3) and weights (24,) summing to 1/6.
    """
    def orbit4(a):
        b = 1.0 - 3.0 * a
        return [(b, a, a), (a, a, a), (a, a, b), (a, b, a)]

    pts = []
    w = []
    for a, wt in ((0.214602871259151684, 0.039922750258167949),
                  (0.040673958534611353, 0.010077211055320643),
                  (0.322337890142275646, 0.055357181543654720)):
        pts += orbit4(a)
        w += [wt] * 4
    a, b, c = 0.063661001875017525, 0.269672331458315867, 0.603005664791649076
    pts += [(b, a, a), (a, b, a), (a, a, b), (c, a, a), (a, c, a), (a, a, c),
            (a, b, c), (b, c, a), (c, a, b), (a, c, b), (b, a, c), (c, b, a)]
    w += [0.048214285714285714] * 12
    return np.array(pts), np.array(w) / 6.0


def triangle12():
    """12-point degree-6 rule on the UFC reference triangle (FIAT `_triangle_scheme(6)`); weights sum 1/2."""
    pts = []
    w = []
    for a, wt in ((0.063089014491502, 0.050844906370207), (0.249286745170910, 0.116786275726379)):
        b = 1.0 - 2.0 * a
        pts += [(a, a), (b, a), (a, b)]
        w += [wt] * 3
    a, b = 0.053145049844817, 0.310352451033784
    c = 1.0 - a - b
    pts += [(a, b), (b, a), (a, c), (c, a), (b, c), (c, b)]
    w += [0.082851075618374] * 6
    return np.array(pts), np.array(w) / 2.0


def tabulate_p2(pts):
    """P2 values (Q,10) and reference gradients (Q,10,3), UFC order (4 vertices, 6 edges)."""
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    lam = np.stack([1.0 - x - y - z, x, y, z], axis=1)                 # (Q,4)
    dlam = np.array([[-1.0, -1.0, -1.0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])  # (4,3)
    N = np.concatenate([lam * (2 * lam - 1), 4 * lam[:, TET_EDGES[:, 0]] * lam[:, TET_EDGES[:, 1]]], axis=1)
    dv = (4 * lam - 1)[:, :, None] * dlam[None]
    de = 4 * (lam[:, TET_EDGES[:, 0], None] * dlam[TET_EDGES[:, 1]][None]
              + lam[:, TET_EDGES[:, 1], None] * dlam[TET_EDGES[:, 0]][None])
    return N, np.concatenate([dv, de], axis=1), lam, dlam


# ------------------------------------------------------------------------------------------------
# small dense 3x3 helpers that stay analytic for complex arguments
# ------------------------------------------------------------------------------------------------

def _det3(A):
    return (A[..., 0, 0] * (A[..., 1, 1] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 1])
            - A[..., 0, 1] * (A[..., 1, 0] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 0])
            + A[..., 0, 2] * (A[..., 1, 0] * A[..., 2, 1] - A[..., 1, 1] * A[..., 2, 0]))


def _inv3(A):
    det = _det3(A)
    c = np.empty_like(A)
    c[..., 0, 0] = A[..., 1, 1] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 1]
    c[..., 0, 1] = A[..., 0, 2] * A[..., 2, 1] - A[..., 0, 1] * A[..., 2, 2]
    c[..., 0, 2] = A[..., 0, 1] * A[..., 1, 2] - A[..., 0, 2] * A[..., 1, 1]
    c[..., 1, 0] = A[..., 1, 2] * A[..., 2, 0] - A[..., 1, 0] * A[..., 2, 2]
    c[..., 1, 1] = A[..., 0, 0] * A[..., 2, 2] - A[..., 0, 2] * A[..., 2, 0]
    c[..., 1, 2] = A[..., 0, 2] * A[..., 1, 0] - A[..., 0, 0] * A[..., 1, 2]
    c[..., 2, 0] = A[..., 1, 0] * A[..., 2, 1] - A[..., 1, 1] * A[..., 2, 0]
    c[..., 2, 1] = A[..., 0, 1] * A[..., 2, 0] - A[..., 0, 0] * A[..., 2, 1]
    c[..., 2, 2] = A[..., 0, 0] * A[..., 1, 1] - A[..., 0, 1] * A[..., 1, 0]
    return c / det[..., None, None], det


I3 = np.eye(3)


class FsiOracle:
    """Assembles F (split into the reference's "linear" and "nonlinear" parts), their Jacobians, applies
    DOLFIN boundary-condition semantics and runs turtleFSI's quasi-Newton policy with a sparse LU.

    ``desc`` is a plain dict of arrays/scalars (no product classes):
      coords (V,3), tets (C,4) vertex-sorted, tet_nodes (C,10) P2 node ids (vertices first),
      num_nodes N2, cell_kind (C,) 0=fluid 1=solid, cell_region (C,) index into fluid_props /
      solid_props, fluid_props [(rho_f, mu_f)], solid_props [(rho_s, mu_s, lambda_s)],
      dt, theta,
      pressure_facets: (nf,6) P2 facet nodes, pressure_facet_cell: (nf,) the '+' (solid) cell,
      robin_facets (nr,6), robin_k (nr,), robin_c (nr,)  (optional),
      bc_dofs (nb,) unique Dirichlet dofs (list order already resolved: later BC wins).
    Global dof layout: [d: 3*N2 | v: 3*N2 | p: V], component-minor.
    """

    def __init__(self, desc, impl: str = "auto"):
        self.D = desc
        self.x = np.asarray(desc["coords"], dtype=float)
        self.tets = np.asarray(desc["tets"])
        self.tn = np.asarray(desc["tet_nodes"])
        self.N2 = int(desc["num_nodes"])
        self.V = len(self.x)
        self.ndof = 6 * self.N2 + self.V
        self.dt = float(desc["dt"])
        self.theta = float(desc["theta"])
        self.kind = np.asarray(desc["cell_kind"])
        self.region = np.asarray(desc["cell_region"])
        # quadrature + tabulation -------------------------------------------------------------
        self.qp, self.qw = keast24()
        self.N, self.dNref, self.L, self.dL = tabulate_p2(self.qp)
        xc = self.x[self.tets]                                          # (C,4,3)
        Jm = np.stack([xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0], xc[:, 3] - xc[:, 0]], axis=2)  # dx/dxi
        Jinv, det = _inv3(Jm)
        self.detJ = det
        self.wdet = np.abs(det)[:, None] * self.qw[None, :]             # (C,Q)
        # physical gradients: dN/dx_j = sum_k dN/dxi_k * dxi_k/dx_j
        self.G = np.einsum("qak,ckj->cqaj", self.dNref, Jinv)           # (C,Q,10,3)
        self.GL = np.einsum("ak,ckj->caj", self.dL, Jinv)               # (C,4,3) P1 gradients (constant)
        # dof map -----------------------------------------------------------------------------
        tn, N2 = self.tn, self.N2
        cols = [off + 3 * tn + c for off in (0, 3 * N2) for c in range(3)]
        cols.append(6 * N2 + self.tets)
        self.cell_dofs = np.concatenate(cols, axis=1)                   # (C,64)
        self.bc_dofs = np.asarray(desc.get("bc_dofs", np.zeros(0, dtype=np.int64)), dtype=np.int64)
        self._facet_setup()
        # the same element arithmetic in C under OpenMP (oracle/fsi_oracle_c.c), used when its library is there;
        # ORACLE_IMPL=numpy forces the numpy definition below (tests/test_oracle_c.py holds the two together)
        self.c = None
        if impl != "numpy":
            from . import c_oracle
            if c_oracle.available():
                self.c = c_oracle.CElements(self)

    # ---- local <-> global -----------------------------------------------------------------------
    def gather(self, U):
        return U[self.cell_dofs]

    @staticmethod
    def unpack(loc):
        """(C,64) -> d (C,10,3), v (C,10,3), p (C,4)"""
        d = np.stack([loc[:, 0:10], loc[:, 10:20], loc[:, 20:30]], axis=2)
        v = np.stack([loc[:, 30:40], loc[:, 40:50], loc[:, 50:60]], axis=2)
        return d, v, loc[:, 60:64]

    @staticmethod
    def pack(rd, rv, rp):
        return np.concatenate([rd[:, :, 0], rd[:, :, 1], rd[:, :, 2], rv[:, :, 0], rv[:, :, 1], rv[:, :, 2], rp], axis=1)

    # ---- element residuals ----------------------------------------------------------------------
    def _kin(self, cells, d, v):
        G = self.G[cells]
        N = self.N
        gd = np.einsum("cai,cqaj->cqij", d, G)
        gv = np.einsum("cai,cqaj->cqij", v, G)
        dq = np.einsum("qa,cai->cqi", N, d)
        vq = np.einsum("qa,cai->cqi", N, v)
        return gd, gv, dq, vq

    def _fluid_residual(self, cells, rho, mu, loc, loc1):
        """turtleFSI fluid.py + laplace.py on fluid cells. Returns (R_linear, R_nonlinear), each (c,64)."""
        k, th0, th1 = self.dt, self.theta, 1.0 - self.theta
        d, v, p = self.unpack(loc)
        d1, v1, _ = self.unpack(loc1)
        G, N, L, w = self.G[cells], self.N, self.L, self.wdet[cells]
        gd, gv, dq, vq = self._kin(cells, d, v)
        gd1, gv1, dq1, vq1 = self._kin(cells, d1, v1)
        pq = np.einsum("qa,ca->cq", L, p)
        Finv, J = _inv3(I3 + gd)
        Finv1, J1 = _inv3(I3 + gd1)
        A = gv @ Finv            # grad(v) * inv(F)
        A1 = gv1 @ Finv1
        FinvT = np.swapaxes(Finv, -1, -2)
        Finv1T = np.swapaxes(Finv1, -1, -2)
        dv = vq - vq1
        # ---------------- nonlinear part (F_fluid_nonlinear) ----------------
        val_n = (rho / k) * (J * th0)[..., None] * dv                                    # temporal
        val_n = val_n + th0 * rho * J[..., None] * np.einsum("cqij,cqj->cqi", A, vq)     # convection
        val_n = val_n - (rho / k) * J[..., None] * np.einsum("cqij,cqj->cqi", A, dq - dq1)  # ALE
        sig = mu * (A + np.swapaxes(A, -1, -2))
        grd_n = J[..., None, None] * (-pq[..., None, None] * FinvT)                      # pressure stress
        grd_n = grd_n + th0 * J[..., None, None] * (sig @ FinvT)                         # viscous stress
        cont = J * np.einsum("cqii->cq", A)          # div(J F^-1 v) = J tr(grad(v) F^-1)  (Piola identity)
        # ---------------- linear part (F_fluid_linear, incl. laplace) ----------------
        val_l = (rho / k) * (J1 * th1)[..., None] * dv
        val_l = val_l + th1 * rho * J1[..., None] * np.einsum("cqij,cqj->cqi", A1, vq1)
        sig1 = mu * (A1 + np.swapaxes(A1, -1, -2))
        grd_l = th1 * J1[..., None, None] * (sig1 @ Finv1T)
        lap = 1.0 * gd                                                                   # alfa = 1 ("constant")

        def test_v(val, grd):
            return (np.einsum("cq,qa,cqi->cai", w, N, val) + np.einsum("cq,cqaj,cqij->cai", w, G, grd))

        zero_d = np.zeros_like(d)
        zero_p = np.zeros_like(p)
        Rn = self.pack(zero_d, test_v(val_n, grd_n), np.einsum("cq,qa,cq->ca", w, L, cont))
        Rl = self.pack(np.einsum("cq,cqaj,cqij->cai", w, G, lap), test_v(val_l, grd_l), zero_p)
        return Rl, Rn

    def _solid_residual(self, cells, rho, mu, lam, loc, loc1, model=0, C10=0.0, C01=0.0, C11=0.0):
        """turtleFSI solid.py with S = lambda tr(E) I + 2 mu E (StVenantKirchoff), or the compressible Mooney-Rivlin
        energy psi = C10 (I1b-3) + C01 (I2b-3) + C11 (I1b-3)(I2b-3) + K (J ln J - J + 1), K = lambda + 2 mu/3
        (isochoric invariants; S = 2 dpsi/dC).  The reference's exact energy lives in the un-vendored turtleFSI and no
        reference test pins a number for it: parity unpinned for this model."""
        k, th0, th1 = self.dt, self.theta, 1.0 - self.theta
        d, v, p = self.unpack(loc)
        d1, v1, _ = self.unpack(loc1)
        G, N, w = self.G[cells], self.N, self.wdet[cells]
        gd, gv, dq, vq = self._kin(cells, d, v)
        gd1, gv1, dq1, vq1 = self._kin(cells, d1, v1)

        def piola(g):
            F = I3 + g
            if model == 1:
                C = np.swapaxes(F, -1, -2) @ F
                Ci, detC = _inv3(C)
                J = _det3(F)
                I1 = np.einsum("cqii->cq", C)
                I2 = 0.5 * (I1 ** 2 - np.einsum("cqij,cqji->cq", C, C))
                Jm23 = J ** (-2.0 / 3.0)
                I1b, I2b = Jm23 * I1, Jm23 ** 2 * I2
                a1 = (2.0 * (C10 + C11 * (I2b - 3.0)) * Jm23)[..., None, None]
                a2 = (2.0 * (C01 + C11 * (I1b - 3.0)) * Jm23 ** 2)[..., None, None]
                K = lam + 2.0 * mu / 3.0
                S = (a1 * (I3 - (I1 / 3.0)[..., None, None] * Ci)
                     + a2 * (I1[..., None, None] * I3 - C - (2.0 / 3.0) * I2[..., None, None] * Ci)
                     + (K * np.log(J) * J)[..., None, None] * Ci)
                return F @ S
            E = 0.5 * (np.swapaxes(F, -1, -2) @ F - I3)
            trE = np.einsum("cqii->cq", E)
            S = lam * trE[..., None, None] * I3 + 2.0 * mu * E
            return F @ S

        val_v = (rho / k) * (vq - vq1)
        val_d = DELTA * rho * (1.0 / k) * (dq - dq1) - DELTA * rho * (th0 * vq + th1 * vq1)
        Rl = self.pack(np.einsum("cq,qa,cqi->cai", w, N, val_d),
                       np.einsum("cq,qa,cqi->cai", w, N, val_v)
                       + np.einsum("cq,cqaj,cqij->cai", w, G, th1 * piola(gd1)),
                       np.zeros_like(p))
        Rn = self.pack(np.zeros_like(d), np.einsum("cq,cqaj,cqij->cai", w, G, th0 * piola(gd)), np.zeros_like(p))
        return Rl, Rn

    def _groups(self):
        D = self.D
        for r, (rho, mu) in enumerate(D["fluid_props"]):
            cells = np.nonzero((self.kind == 0) & (self.region == r))[0]
            if len(cells):
                yield cells, (lambda c, a, b, rho=rho, mu=mu: self._fluid_residual(c, rho, mu, a, b))
        models = D.get("solid_models", [0] * len(D["solid_props"]))
        for r, props in enumerate(D["solid_props"]):
            rho, mu, lam = props[:3]
            extra = dict(model=int(models[r]), C10=props[3], C01=props[4], C11=props[5]) if len(props) >= 6 else {}
            cells = np.nonzero((self.kind == 1) & (self.region == r))[0]
            if len(cells):
                yield cells, (lambda c, a, b, rho=rho, mu=mu, lam=lam, extra=extra:
                              self._solid_residual(c, rho, mu, lam, a, b, **extra))

    def element_residuals(self, U, U1):
        """(R_linear, R_nonlinear): per-element vectors (C,64)."""
        if self.c is not None and not np.iscomplexobj(U) and not np.iscomplexobj(U1):
            return self.c.residuals(np.asarray(U, dtype=float), np.asarray(U1, dtype=float))
        loc, loc1 = self.gather(U), self.gather(U1)
        Rl = np.zeros(loc.shape, dtype=loc.dtype)
        Rn = np.zeros(loc.shape, dtype=loc.dtype)
        for cells, fn in self._groups():
            a, b = fn(cells, loc[cells], loc1[cells])
            Rl[cells], Rn[cells] = a, b
        return Rl, Rn

    def element_jacobians(self, U, U1):
        """(J_linear, J_nonlinear) element matrices (C,64,64) = d R_e / d U^n_e by complex step."""
        if self.c is not None:
            return self.c.jacobians(np.asarray(U, dtype=float), np.asarray(U1, dtype=float))
        h = 1e-30
        loc, loc1 = self.gather(U).astype(complex), self.gather(U1).astype(complex)
        C = len(loc)
        Jl = np.zeros((C, 64, 64))
        Jn = np.zeros((C, 64, 64))
        for cells, fn in self._groups():
            a, b = loc[cells], loc1[cells]
            for j in range(64):
                ap = a.copy()
                ap[:, j] += 1j * h
                rl, rn = fn(cells, ap, b)
                Jl[cells, :, j] = rl.imag / h
                Jn[cells, :, j] = rn.imag / h
        return Jl, Jn

    # ---- facet terms ----------------------------------------------------------------------------
    def _facet_setup(self):
        D = self.D
        tp, tw = triangle12()
        l = np.stack([1 - tp[:, 0] - tp[:, 1], tp[:, 0], tp[:, 1]], axis=1)
        TRI_E = np.array([[1, 2], [0, 2], [0, 1]])
        Nf = np.concatenate([l * (2 * l - 1), 4 * l[:, TRI_E[:, 0]] * l[:, TRI_E[:, 1]]], axis=1)   # (12,6)
        self.facet_mass_w = np.einsum("q,qa->a", tw, Nf) * 2.0          # ∫N_a / area
        self.facet_mass = np.einsum("q,qa,qb->ab", tw, Nf, Nf) * 2.0     # ∫N_a N_b / area
        pf = np.asarray(D.get("pressure_facets", np.zeros((0, 6), dtype=np.int64)))
        self.pf = pf
        if len(pf):
            x = self.x
            f = pf[:, :3]
            nv = np.cross(x[f[:, 1]] - x[f[:, 0]], x[f[:, 2]] - x[f[:, 0]])     # |nv| = 2 area
            cc = x[self.tets[np.asarray(D["pressure_facet_cell"])]].mean(axis=1)
            flip = np.einsum("ij,ij->i", nv, x[f].mean(axis=1) - cc) < 0
            nv[flip] *= -1                                                       # outward from the '+' cell
            self.pf_nA = 0.5 * nv                                                # n * area
        rf = np.asarray(D.get("robin_facets", np.zeros((0, 6), dtype=np.int64)))
        self.rf = rf
        if len(rf):
            x = self.x
            f = rf[:, :3]
            self.rf_area = 0.5 * np.linalg.norm(np.cross(x[f[:, 1]] - x[f[:, 0]], x[f[:, 2]] - x[f[:, 0]]), axis=1)
            self.rf_k = np.asarray(D["robin_k"], dtype=float)
            self.rf_c = np.asarray(D["robin_c"], dtype=float)

    def facet_residual(self, U, U1, P):
        """F_solid_linear facet terms: P(t) (n('+'), psi('+')) dS(fsi) [REF offset_stenosis.py:190]
        and the Robin term theta0 (k_s d + c_s v, psi) ds + theta1 (...old...) [turtleFSI solid.py]."""
        R = np.zeros(self.ndof, dtype=np.result_type(U.dtype, float))
        N2 = self.N2
        if len(self.pf):
            contrib = P * self.pf_nA[:, None, :] * self.facet_mass_w[None, :, None]   # (nf,6,3)
            dofs = 3 * N2 + 3 * self.pf[:, :, None] + np.arange(3)[None, None, :]
            np.add.at(R, dofs.ravel(), contrib.ravel())
        if len(self.rf):
            th0, th1 = self.theta, 1 - self.theta
            dn = U[(3 * self.rf[:, :, None] + np.arange(3)).reshape(len(self.rf), 6, 3)]
            vn = U[3 * N2 + (3 * self.rf[:, :, None] + np.arange(3))]
            d1 = U1[(3 * self.rf[:, :, None] + np.arange(3))]
            v1 = U1[3 * N2 + (3 * self.rf[:, :, None] + np.arange(3))]
            q = (th0 * (self.rf_k[:, None, None] * dn + self.rf_c[:, None, None] * vn)
                 + th1 * (self.rf_k[:, None, None] * d1 + self.rf_c[:, None, None] * v1))
            contrib = self.rf_area[:, None, None] * np.einsum("ab,fbi->fai", self.facet_mass, q)
            dofs = 3 * N2 + 3 * self.rf[:, :, None] + np.arange(3)[None, None, :]
            np.add.at(R, dofs.ravel(), contrib.ravel())
        return R

    def facet_jacobian(self):
        """d(facet_residual)/dU^n (Robin only; the pressure term does not depend on U)."""
        n = self.ndof
        if not len(self.rf):
            return sp.csr_matrix((n, n))
        N2, th0 = self.N2, self.theta
        rows, cols, vals = [], [], []
        M = self.rf_area[:, None, None] * self.facet_mass[None]        # (nr,6,6)
        for i in range(3):
            r = 3 * N2 + 3 * self.rf + i
            for off, coef in ((0, self.rf_k), (3 * N2, self.rf_c)):
                c = off + 3 * self.rf + i
                rows.append(np.repeat(r[:, :, None], 6, axis=2).ravel())
                cols.append(np.repeat(c[:, None, :], 6, axis=1).ravel())
                vals.append((th0 * coef[:, None, None] * M).ravel())
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))

    # ---- global assembly ------------------------------------------------------------------------
    def assemble_vector(self, Re):
        return np.bincount(self.cell_dofs.ravel(), weights=Re.ravel(), minlength=self.ndof)

    def assemble_matrix(self, Je):
        if self.c is not None:
            return self.c.assemble_matrix(Je)
        rows = np.repeat(self.cell_dofs[:, :, None], 64, axis=2).ravel()
        cols = np.repeat(self.cell_dofs[:, None, :], 64, axis=1).ravel()
        A = sp.coo_matrix((Je.ravel(), (rows, cols)), shape=(self.ndof, self.ndof)).tocsr()
        return A

    def residual(self, U, U1, P=0.0):
        """Assembled F(U^n; U^{n-1}) (not negated, no BCs)."""
        Rl, Rn = self.element_residuals(U, U1)
        return self.assemble_vector(Rl + Rn) + self.facet_residual(U, U1, P)

    def function_norm(self, X):
        """``norm(dvp_res, 'l2')`` of the reference's newtonsolver.  ``dolfin.norm`` of a *Function* (not a vector)
        assembles sqrt(int_Omega |f|^2 dx) whatever the case of the norm name, so "r (rel)" in the log is the
        L2(Omega) norm of the mixed update (d, v, p), not the l2 norm of its dof vector.  Cross-check: the tutorial
        log [REF docs/offset_stenosis.md:200-202] shows r (rel) = 2e-2 at iteration 0 where the dof-vector norm of a
        pressure update of O(10 Pa) over O(10^3) vertices would be O(10^2)."""
        d, v, p = self.unpack(self.gather(X))
        w, N, L = self.wdet, self.N, self.L
        tot = 0.0
        for fld in (d, v):
            fq = np.einsum("qa,cai->cqi", N, fld)
            tot += np.einsum("cq,cqi,cqi->", w, fq, fq)
        pq = np.einsum("qa,ca->cq", L, p)
        tot += np.einsum("cq,cq,cq->", w, pq, pq)
        return float(np.sqrt(tot))

    # ---- turtleFSI newtonsolver.py ----------------------------------------------------------------
    def solver_setup(self, U, U1):
        """``A_pre = assemble(J_linear)`` once, at the state the solver is created with."""
        Jl, _ = self.element_jacobians(U, U1)
        self.A_pre = self.assemble_matrix(Jl) + self.facet_jacobian()
        self.lu = None

    def jacobian(self, U, U1):
        """A = assemble(J_nonlinear) + A_pre; ident_zeros; bc.apply(A)."""
        _, Jn = self.element_jacobians(U, U1)
        A = (self.assemble_matrix(Jn) + self.A_pre).tocsr()
        A.eliminate_zeros()
        # ident_zeros: rows without any non-zero entry get a unit diagonal
        rownnz = np.diff(A.indptr)
        zero_rows = np.nonzero(rownnz == 0)[0]
        # bc.apply(A): zero the row, one on the diagonal
        keep = np.ones(self.ndof)
        keep[self.bc_dofs] = 0.0
        A = sp.diags(keep) @ A
        diag = np.zeros(self.ndof)
        diag[self.bc_dofs] = 1.0
        diag[zero_rows] = 1.0
        A = (A + sp.diags(diag)).tocsr()
        self.zero_rows = zero_rows
        return A

    def rhs(self, U, U1, P, bc_values):
        """b = assemble(-F); bc.apply(b, u): b[dof] = g - u[dof]."""
        b = -self.residual(U, U1, P)
        b[self.bc_dofs] = bc_values - U[self.bc_dofs]
        return b

    def newtonsolver(self, U, U1, P, bc_values, *, atol, rtol, max_it, lmbda, recompute, recompute_tstep,
                     counter, first_step_num, log=None):
        """One time step of turtleFSI's quasi-Newton loop. Updates U in place.

        Returns the list of (residual, rel_res, recomputed) per iteration.
        """
        it = 0
        residual = 1e8
        rel_res = 1e8
        last_residual = residual
        hist = []
        while rel_res > rtol and residual > atol and it < max_it:
            rec = ((it == 0 and counter % recompute_tstep == 0)
                   or (it > 0 and it % recompute == 0)
                   or (it > 0 and last_residual < residual)
                   or (it == 0 and counter == first_step_num))
            if rec or self.lu is None:
                if log:
                    log("Compute Jacobian matrix")
                A = self.jacobian(U, U1)
                self.A = A
                self.lu = spla.splu(A.tocsc())
            b = self.rhs(U, U1, P, bc_values)
            last_residual = residual
            dU = self.lu.solve(b)
            U += lmbda * dU
            U[self.bc_dofs] = bc_values
            residual = float(np.linalg.norm(b))
            rel_res = self.function_norm(dU)
            if not np.isfinite(residual) or not np.isfinite(rel_res) or residual > 1e20 or rel_res > 1e20:
                raise RuntimeError("Error: The simulation has diverged during the Newton solve.")
            if log:
                log("Newton iteration %d: r (atol) = %.3e (tol = %.3e), r (rel) = %.3e (tol = %.3e) "
                    % (it, residual, atol, rel_res, rtol))
            hist.append((residual, rel_res, bool(rec)))
            it += 1
        return hist
