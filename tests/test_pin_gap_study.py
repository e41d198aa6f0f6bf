"""Where the oracle stands against the reference's known answers, as checks instead of prose.

The reference cannot run here, so what can be established about the remaining gap between the oracle and
the pinned numbers of REF tests/test_create_hdf5_and_separate_viz.py:40-51,196-206 / tests/test_predeform.py:32-33 is
established on the cylinder case (1 647 tets, 3 steps), where every step is cheap enough to perturb:

1. OUR linear solves are not the source: one SuperLU solve of the first Newton system is accurate to 1e-10 of v_x (the
   matrix is well-conditioned componentwise but has cond_1 = 4e15: what the reference's MUMPS returned for it is another matter);
2. the structure of the discrete equations is right where the pins can see it: the pinned pairs (v_x, d_x) at the interface
   vertex violate d = dt (theta v^n + (1 - theta) v^{n-1}) by 2.695e-5 d - the footprint of the fluid's Laplace lifting
   term (alfa = 1) on a d-row whose penalty is delta rho_s / k - and the oracle reproduces that number to five digits at all
   three steps (delta = 1e7, alfa, theta, the P2 mass and stiffness matrices are therefore the reference's);
3. stopping noise is not the source either: after the first Newton iteration from rest (exact Jacobian) v_x is 6.1e-5 above
   the pin, after the second it has converged and is 2.1e-5 below; no iteration count of the restated algorithm lands on
   the pin, so the difference is in the discrete equations or their data, at the 2e-5 level (8e-5 at step 2);
4. it is not one of the obvious candidates: no single physical parameter (rho_f, mu_f, rho_s, mu_s, lambda_s, the load, the
   inlet amplitude, delta) moved within a linear fit closes the nine gaps.  (Also tried in round 2, not part of this
   file: the docs' Nanson-formula variant of the interface load, REF docs/aneurysm.md:126-135, moves step 1 from -2.1e-5
   to -2.7e-5, i.e. away from the pin; rho_f = 1000; partial-nonlinearity variants of the solid stress.)

5. where the sensitivity lives (round 2, late): the step from the first Newton iterate (the linear solution, +6.1e-5 above the
   pin) to the converged state (-2.1e-5 below) is the SOLID's geometric nonlinearity and nothing else - with the solid
   linearised the converged v_x stays at +6.0e-5, with S(E) but without the factor F in P = F S at +4.6e-5, with F S(eps)
   (linear strain, full F) at -6.5e-6, while switching off every F / J factor of the FLUID residual together moves v_x by
   1.1e-6.  The pin sits at 0.74 of the solid's quadratic term - which is NOT a hint that the reference weights that term
   differently: on the offset-stenosis known-answer case, where the solid's nonlinearity is an 8 % effect, a weight of
   0.744 moves v_x by 2e-2, so the 2.6e-5 gap there pins the weight to 1 +- 3e-4 (run of 2026-10: numpy oracle, converged).
   The gap is a ~2e-5 effect at the LINEAR level in both problems, whatever the strength of the nonlinearity.  With the
   reference's own stopping rule the oracle stops
   after ONE iteration in step 1 (update norm below 1e-6), i.e. at +6.1e-5; the known answer is neither that iterate nor
   the converged one.  Also ruled out: the degree-4 quadrature that turtleFSI's default `compiler_parameters` give the
   Jacobian of the cylinder and predeform problems (their problem files do not pass `compiler_parameters`, REF
   cylinder.py:21 vs offset_stenosis.py:78) - same iterates to 1e-8; the `last_rel_res` / never variants of the
   recompute-on-increase rule on the stenosis case (p gap 3.5e-4 .. 4.7e-4 against 3.8e-4).

6. what the gap looks like as data (round 2, last study): a factor on the interface load of 1 + 2.144e-5, 1 + 4.163e-5,
   1 + 5.676e-5 at steps 1, 2, 3 lands d_x on its three pins exactly and then v_x on its pins to 1.3e-9 relative (v follows
   from the d history through the penalty row), but the same run leaves d_y and d_z of the predeform pins 9.5e-5 and 4.2e-5
   off (unfitted: -6.0e-5, +3.0e-5, -1.8e-5 for x, y, z): the difference is a field, not a scale factor of the response
   (the tube's axis is y, vertex 0 sits on the interface at mid-length: at step 3 the gap is -1.9e-5 of the radial displacement plus a
   circumferential part 45 % as large, where the displacement itself is 3.8 % circumferential - every global change of a
   parameter, the load or a weight of the old / new stress term moves x, y, z alike, so the difference looks local and
   mesh-dependent, like the Laplace footprint of item 2).  In
   load units it grows like 1 : 7.76 : 23.8 over the three steps - faster than the load (1 : 4 : 9) or the displacement
   (1 : 3.16 : 6.16) - and no term of the restated equations has that size and shape: convection and the ALE term together
   move d_x by 5e-7 .. 5e-6, the four quadratic pieces of the solid stress (lambda tr(eps) g, 2 mu g eps, lambda/2 |g|^2 I,
   mu g^T g) by -7.0e-5, +3.4e-6, -7.7e-6, -6.3e-6 at step 1 with ratios to the gap that change from step to step.  With the
   reference's stopping rule every step takes exactly two iterations with the Jacobian at rest (|b| = 7.8e-4, 5.3e-10 at
   step 1), which is converged to 1e-7: the reference solves ITS equations as tightly as the oracle solves these.

7. round 3, two more candidates with numbers (both as tests below).  (a) FFC's uflacs clamps table entries that are close to
   -1, -1/2, 0, 1/2, 1 ("table_rtol" 1e-6, "table_atol" 1e-9 in FFC 2019.x; numpy's 1e-5 / 1e-8 if a version passed none): the
   P2 tables of the 24-point cell rule keep 6.6e-2 away from +-1/2 and 7.9e-2 from +-1, but on the 12-point FACET rule the
   edge functions take the value 4 a (1 - 2a) = 0.4999959 at a = 0.2492867..., 4.07e-6 from 1/2: inside numpy's default
   tolerance (5.0e-6), outside FFC's (5.0e-7).  If it were clamped the interface load would grow by 2.85e-6 uniformly - a
   seventh of the gap and of the wrong shape (a load factor moves x, y, z alike, item 6).  (b) a partial second Newton
   correction: the pin of step 1 sits at 0.748 of the way from the first (linear) iterate to the converged state, and a
   factor 0.748 on the second correction of EVERY step brings d_x to 1.2e-8 / 8.6e-6 / 9.8e-6 and v_x to 1.2e-8 / 2.3e-5 /
   3.9e-6 of their pins (from 2.1e-5 .. 8.7e-5) - but moves the axial displacement at step 3 from 3.1e-5 to 1.2e-4 off: like
   the load factor of item 6 it fits the radial response and no more, and nothing in the restated algorithm scales a
   correction (lmbda = 1; with the Jacobian at rest exact, two iterations are converged to 5e-8).

What follows from 1-7 for the parity status is written in DESIGN.md §2; the strict-xfail tests at the end hold the
reference's own tolerances and will flip the day the gap is closed.
"""
import contextlib
import io

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from conftest import GOLDEN

PIN_V = np.array([4.38261949610407e-6, 5.244315455211961e-6, 8.137814761280497e-6])
PIN_D = np.array([2.235075700301419e-9, 7.0569699656660426e-9, 1.3776599148439903e-8])
PIN_PRE = np.array([7.382372340085156e-5, -1.1083576098054155e-4, 4.930899508039441e-4])
DT, THETA = 1e-3, 0.51


@pytest.fixture(scope="module")
def study(cylinder_case):
    """The cylinder problem with the Jacobian at rest factorised once; run(policy) replays the three steps."""
    from oracle.fsi_oracle import FsiOracle
    ns, desc, bc_values, pressure, hook = cylinder_case
    o = FsiOracle(desc)
    Z = np.zeros(o.ndof)
    o.solver_setup(Z, Z)
    A = o.jacobian(Z, Z)
    lu = spla.splu(A.tocsc())
    N2 = o.N2

    def data(k):
        with contextlib.redirect_stdout(io.StringIO()):
            ns["t"] = DT * (k + 1)
            hook("pre_solve")(**ns)
        return bc_values(), float(pressure.P)

    def run(max_newton, oracle=o, pscale=1.0, vscale=1.0):
        U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
        vs, ds = [], []
        for k in range(3):
            g, P = data(k)
            g, P = g * vscale, P * pscale
            for it in range(max_newton[k]):
                dU = lu.solve(oracle.rhs(U, U1, P, g))
                U += dU
                U[o.bc_dofs] = g
            vs.append(U[3 * N2]); ds.append(U[0])
            U1[:] = U
        return np.array(vs), np.array(ds), U[:3].copy()

    return dict(o=o, A=A, lu=lu, run=run, data=data, ns=ns, desc=desc)


def test_linear_solve_is_accurate(study):
    o, A, lu = study["o"], study["A"], study["lu"]
    g, P = study["data"](0)
    Z = np.zeros(o.ndof)
    b = o.rhs(Z, Z, P, g)
    x = lu.solve(b)
    r = b - A @ x
    dx = lu.solve(r)                                    # one step of iterative refinement
    assert abs(dx[3 * o.N2]) <= 1e-10 * abs(x[3 * o.N2])


def test_the_newton_matrix_is_normwise_singular_to_double_precision(study):
    """cond_1 of the first Newton matrix is ~4e15 (rows of the delta = 1e7 penalty next to rows of size 1e-9), while entrywise
    perturbations of a few ulp move the pinned velocity by 1e-10: SuperLU's answer (item 1) is accurate, but a factorisation
    with other scaling / pivoting choices (the reference's MUMPS) is only bounded by cond * eps on such a matrix."""
    o, A, lu = study["o"], study["A"].tocsc(), study["lu"]
    inv = spla.LinearOperator(A.shape, matvec=lu.solve, rmatvec=lambda x: lu.solve(x, trans="T"))
    cond = spla.onenormest(A) * spla.onenormest(inv)
    assert cond > 1e15
    rowmax = np.asarray(abs(A).max(axis=1).todense()).ravel()
    assert rowmax.min() < 1e-8 and rowmax.max() > 1e3
    g, P = study["data"](0)
    Z = np.zeros(o.ndof)
    b = o.rhs(Z, Z, P, g)
    x0 = lu.solve(b)
    rng = np.random.default_rng(0)
    Ap = A.copy()
    Ap.data = Ap.data * (1.0 + 4.4e-16 * rng.standard_normal(Ap.nnz))
    xp = spla.splu(Ap).solve(b)
    assert abs(xp[3 * o.N2] / x0[3 * o.N2] - 1) < 1e-9


def test_pins_carry_the_laplace_footprint_and_the_oracle_reproduces_it(study):
    def violation(v, d):
        vbar = np.array([THETA * v[0], THETA * v[1] + (1 - THETA) * v[0], THETA * v[2] + (1 - THETA) * v[1]])
        inc = np.array([d[0], d[1] - d[0], d[2] - d[1]])
        return (DT * vbar - inc) / d
    pin = violation(PIN_V, PIN_D)
    v, d, _ = study["run"]([3, 3, 3])
    ours = violation(v, d)
    assert np.allclose(pin, 2.695e-5, rtol=2e-3)         # the pins themselves: 2.6953e-5, 2.6948e-5, 2.6924e-5
    assert np.abs(ours / pin - 1).max() < 5e-5           # same number from the restated equations, five digits


def test_no_iteration_count_lands_on_the_pin(study):
    v1, _, _ = study["run"]([1, 0, 0])
    v2, _, _ = study["run"]([2, 0, 0])
    v3, _, _ = study["run"]([3, 0, 0])
    assert abs(v3[0] / v2[0] - 1) < 1e-7                                  # two iterations have converged (to 5e-8)
    above, below = v1[0] / PIN_V[0] - 1, v2[0] / PIN_V[0] - 1
    assert 5.5e-5 < above < 6.7e-5 and -2.4e-5 < below < -1.8e-5            # +6.1e-5 / -2.1e-5: the pin lies strictly between


def test_no_single_parameter_closes_the_gaps(study):
    import copy
    from oracle.fsi_oracle import FsiOracle
    import oracle.fsi_oracle as fo
    X0 = study["ns"]["mesh"].coords[0]

    def obs(res):
        v, d, d3 = res
        return np.concatenate([v / PIN_V - 1, d / PIN_D - 1, (d3 - (X0 - PIN_PRE)) / d3])

    run, desc = study["run"], study["desc"]
    y0 = obs(run([3, 4, 4]))
    assert 1.4e-4 < np.linalg.norm(y0) < 1.7e-4          # the nine gaps: 2e-5 ... 8.6e-5 each
    eps = 1e-3
    cols = []

    def variant(key, idx):
        d2 = copy.deepcopy(desc)
        rows = [list(p) for p in d2[key]]
        for p in rows:
            p[idx] *= 1 + eps
        d2[key] = [tuple(p) for p in rows]
        return FsiOracle(d2)

    for key, idx in (("fluid_props", 0), ("fluid_props", 1), ("solid_props", 0), ("solid_props", 1), ("solid_props", 2)):
        cols.append((obs(run([4, 5, 5], oracle=variant(key, idx))) - y0) / eps)
    cols.append((obs(run([3, 4, 4], pscale=1 + eps)) - y0) / eps)
    cols.append((obs(run([3, 4, 4], vscale=1 + eps)) - y0) / eps)
    fo.DELTA = 1e7 * (1 + eps)
    try:
        cols.append((obs(run([4, 5, 5])) - y0) / eps)
    finally:
        fo.DELTA = 1e7
    for c in cols:                                        # best single-parameter fit leaves more than 60 % of the gap
        a = -(c @ y0) / (c @ c)
        assert np.linalg.norm(y0 + a * c) > 0.6 * np.linalg.norm(y0)


def test_the_gap_moves_with_the_solid_nonlinearity_only(cylinder_case):
    """Item 5 of the module docstring, as numbers: converged v_x(vertex 0, t = 1 ms) / pin - 1 for variants of the solid's
    first Piola-Kirchhoff stress (numpy oracle, Jacobian at rest re-used: five iterations converge to 1e-9)."""
    import oracle.fsi_oracle as fo
    from oracle.fsi_oracle import DELTA, I3, FsiOracle
    ns, desc, bc_values, pressure, hook = cylinder_case

    class Variant(FsiOracle):
        mode = "full"

        def _solid_residual(self, cells, rho, mu, lam, loc, loc1, **_):
            k, th0, th1 = self.dt, self.theta, 1.0 - self.theta
            d, v, p = self.unpack(loc)
            d1, v1, _p = self.unpack(loc1)
            G, N, w = self.G[cells], self.N, self.wdet[cells]
            gd, gv, dq, vq = self._kin(cells, d, v)
            gd1, gv1, dq1, vq1 = self._kin(cells, d1, v1)
            m = self.mode

            def piola(g):
                F = I3 + g
                E = 0.5 * (g + np.swapaxes(g, -1, -2)) if m in ("linear", "F_S_eps") else 0.5 * (np.swapaxes(F, -1, -2) @ F - I3)
                S = lam * np.einsum("cqii->cq", E)[..., None, None] * I3 + 2.0 * mu * E
                return S if m in ("linear", "S_E") else F @ S

            val_v = (rho / k) * (vq - vq1)
            val_d = DELTA * rho * (1.0 / k) * (dq - dq1) - DELTA * rho * (th0 * vq + th1 * vq1)
            Rl = self.pack(np.einsum("cq,qa,cqi->cai", w, N, val_d),
                           np.einsum("cq,qa,cqi->cai", w, N, val_v) + np.einsum("cq,cqaj,cqij->cai", w, G, th1 * piola(gd1)),
                           np.zeros_like(p))
            Rn = self.pack(np.zeros_like(d), np.einsum("cq,cqaj,cqij->cai", w, G, th0 * piola(gd)), np.zeros_like(p))
            return Rl, Rn

    o = Variant(desc, impl="numpy")
    Z = np.zeros(o.ndof)
    o.solver_setup(Z, Z)
    lu = spla.splu(o.jacobian(Z, Z).tocsc())
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = DT
        hook("pre_solve")(**ns)
    g, P = bc_values(), float(pressure.P)

    def converged(mode):
        Variant.mode = mode
        U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
        first = None
        for it in range(5):
            U += lu.solve(o.rhs(U, U1, P, g))
            U[o.bc_dofs] = g
            first = U[3 * o.N2] if first is None else first
        return first / PIN_V[0] - 1, U[3 * o.N2] / PIN_V[0] - 1

    first, full = converged("full")
    assert 5.9e-5 < first < 6.4e-5 and -2.3e-5 < full < -1.9e-5             # +6.1e-5 after one iteration, -2.1e-5 converged
    lin = converged("linear")[1]
    s_e = converged("S_E")[1]
    f_eps = converged("F_S_eps")[1]
    assert 5.7e-5 < lin < 6.2e-5 and 4.3e-5 < s_e < 4.8e-5 and -8e-6 < f_eps < -5e-6
    frac = lin / (lin - full)                                                # share of the quadratic term that would hit the pin
    assert 0.72 < frac < 0.77


def test_a_factor_on_the_interface_load_fits_the_x_pins_but_not_the_other_components(study):
    """Item 6 of the module docstring: per-step load factors that land d_x on its pins, what they do to v_x and to the
    y and z components pinned by REF tests/test_predeform.py:32-33."""
    o, lu, data = study["o"], study["lu"], study["data"]
    X0 = study["ns"]["mesh"].coords[0]

    def step(U1, k, m):
        U = U1.copy()
        g, P = data(k)
        for _ in range(4):
            U += lu.solve(o.rhs(U, U1, P * m, g))
            U[o.bc_dofs] = g
        return U

    U1, sigma = np.zeros(o.ndof), []
    for k in range(3):
        m0, m1 = 1.0, 1.0001
        f0, f1 = (step(U1, k, m)[0] / PIN_D[k] - 1 for m in (m0, m1))
        for _ in range(2):                                   # the response is linear in the factor: two secant steps
            m0, f0, m1 = m1, f1, m1 - f1 * (m1 - m0) / (f1 - f0)
            U = step(U1, k, m1)
            f1 = U[0] / PIN_D[k] - 1
        assert abs(f1) < 1e-11 and abs(U[3 * o.N2] / PIN_V[k] - 1) < 3e-9
        sigma.append(m1 - 1)
        U1 = U
    assert np.allclose(sigma, [2.144e-5, 4.163e-5, 5.676e-5], rtol=2e-3)
    other = U1[1:3] / (X0 - PIN_PRE)[1:3] - 1
    assert 8.5e-5 < other[0] < 1.05e-4 and 3.7e-5 < other[1] < 4.7e-5    # y, z: further off than without the factors


def test_even_a_single_precision_factorisation_does_not_leave_the_gap(study):
    """Under the reference's stopping rule (absolute l2 norms, atol = 1e-6 above the force part of |b|) every step of this
    case ends after its second iteration, so an error eps of the first linear solve survives as ~eps^2.  The gap would need
    eps = 4.5e-3; an equilibrated LU in SINGLE precision has eps = 2.7e-4 at the pinned dofs and leaves 3e-8 after the
    second iteration, a double-precision one (any variant, test_linear_solve_is_accurate) nothing: not solver noise."""
    import scipy.sparse as sp
    o, A, lu, data = study["o"], study["A"].tocsc(), study["lu"], study["data"]
    r = 1.0 / np.sqrt(np.abs(A).max(axis=1).toarray().ravel())
    Ar = sp.diags(r) @ A
    c = 1.0 / np.abs(Ar).max(axis=0).toarray().ravel()
    lu32 = spla.splu((Ar @ sp.diags(c)).tocsc().astype(np.float32))

    def solve32(b):
        return c * lu32.solve((r * b).astype(np.float32)).astype(np.float64)

    g, P = data(0)
    Z = np.zeros(o.ndof)
    b0 = o.rhs(Z, Z, P, g)
    x64, x32 = lu.solve(b0), solve32(b0)
    eps = abs(x32[0] / x64[0] - 1)
    assert 5e-5 < eps < 1.5e-3                                              # 2.7e-4: a poor solve, still 17x better than needed

    def two_iterations(solve):
        U = Z.copy()
        for _ in range(2):
            U += solve(o.rhs(U, Z, P, g))
            U[o.bc_dofs] = g
        return U[0] / PIN_D[0] - 1

    exact, single = two_iterations(lu.solve), two_iterations(solve32)
    assert abs(single - exact) < 2e-7 and -2.2e-5 < exact < -1.9e-5         # the second iteration has removed eps


def test_facet_tables_come_within_numpys_but_not_ffcs_tolerance_of_one_half():
    """Item 7a: the closest approach of a P2 table entry to the numbers uflacs clamps to, on the cell and the facet rule."""
    from oracle.fsi_oracle import keast24, triangle12, tabulate_p2
    N, dN, L, _ = tabulate_p2(keast24()[0])
    for T in (N, dN, L):
        T = T[np.abs(T) > 1e-12]                                           # structural zeros of the gradients aside
        assert min(np.abs(T - n).min() for n in (-1.0, -0.5, 0.5, 1.0)) > 6e-2
    tp, tw = triangle12()
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    P = np.array([(1 - a - b) * verts[0] + a * verts[1] + b * verts[2] for a, b in tp])      # facet z = 0 of the reference cell
    Nf = tabulate_p2(P)[0]
    gap = np.abs(Nf - 0.5).min()
    assert 4.0e-6 < gap < 4.1e-6                                           # 4 a (1 - 2a) at a = 0.249286745170910
    assert gap < 1e-8 + 1e-5 * 0.5 and gap > 1e-9 + 1e-6 * 0.5             # inside numpy's defaults, outside FFC's table_rtol / atol
    # clamped, every edge function would gain 2 points x weight x gap: the load changes by that over 1/3
    hit = np.abs(Nf - 0.5) < 5e-6
    dload = (tw[:, None] * hit * (0.5 - Nf)).sum(axis=0) / 0.5 / (1.0 / 3.0)
    on_face = [6, 8, 9]                                                    # edges (1,2), (0,2), (0,1) of the face z = 0
    assert np.allclose(dload[on_face], 2.85e-6, rtol=2e-2) and np.allclose(np.delete(dload, on_face), 0.0)


def test_a_partial_second_correction_fits_the_x_pins_but_not_the_axial_displacement(study):
    """Item 7b."""
    o, lu, data = study["o"], study["lu"], study["data"]
    N2 = o.N2
    X0 = study["ns"]["mesh"].coords[0]

    def run(f):
        U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
        vs, ds = [], []
        for k in range(3):
            g, P = data(k)
            for scale in (1.0, f):
                U += scale * lu.solve(o.rhs(U, U1, P, g))
                U[o.bc_dofs] = g
            vs.append(U[3 * N2]); ds.append(U[0])
            U1[:] = U
        return np.array(vs), np.array(ds), U[:3].copy()

    v1, d1, e1 = run(1.0)
    v, d, e = run(0.748)
    assert np.abs(d1 / PIN_D - 1).max() > 6e-5 and np.abs(d / PIN_D - 1).max() < 1.1e-5
    assert np.abs(v1 / PIN_V - 1).max() > 8e-5 and np.abs(v / PIN_V - 1).max() < 2.5e-5
    axial = lambda dd: abs(((X0 - dd) - PIN_PRE)[1] / dd[1])
    assert axial(e1) < 3.5e-5 and axial(e) > 1.1e-4                        # the fit of x pays with y


@pytest.mark.xfail(strict=True, reason="oracle vs reference pin: 4.5e-10 / 5.7e-10 against the reference's atol 1e-10 (+ rtol 1e-5); "
                                       "see the module docstring and DESIGN.md §2")
def test_reference_tolerance_on_cylinder_velocity_pins():
    """REF tests/test_create_hdf5_and_separate_viz.py:40-46,196-201, verbatim tolerance."""
    S = np.load(GOLDEN / "cylinder_tight.npz")["states"]
    N2 = (S.shape[1] - 352) // 6
    assert np.isclose(S[:, 3 * N2], PIN_V, atol=1e-10).all()


def test_reference_tolerance_holds_for_the_displacement_pins_and_the_first_velocity_pin():
    """REF tests/test_create_hdf5_and_separate_viz.py:47-51,202-206 and tests/test_predeform.py:32-33, verbatim tolerances."""
    S = np.load(GOLDEN / "cylinder_tight.npz")["states"]
    N2 = (S.shape[1] - 352) // 6
    assert np.isclose(S[:, 0], PIN_D, atol=1e-10).all()
    assert np.isclose(S[0, 3 * N2], PIN_V[0], atol=1e-10)
    from vasp_amd.mesh import FsiMesh
    x0 = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5").coords[0]
    assert np.allclose(x0 - S[2, :3], PIN_PRE, atol=1e-10)


# ---- item 10 (round 5, the last bounded attempt): ALL knobs at once ---------------------------------------------------------
def _pin_fit():
    z = np.load(GOLDEN / "pin_fit.npz")
    y = np.concatenate([z["cylinder_y0"], z["stenosis_y0"]])               # (value - pin) / (atol + 1e-5 |pin|): |.| <= 1 passes the reference
    A = np.vstack([z["cylinder_cols"], z["stenosis_cols"]])
    return [str(p) for p in z["params"]], y, A, float(z["cylinder_linearity"]), float(z["stenosis_linearity"])


def test_joint_fit_of_all_parameters_and_time_level_weights():
    """VERDICT r4 item 1.  tests/golden/make_pin_fit.py differentiated the SIXTEEN pinned numbers (cylinder: v_x, d_x at vertex 0
    for three steps + the predeformed vertex; stenosis after five steps: probe velocity, pressure, solid-probe displacement) with
    respect to thirteen knobs moved one at a time around the restated equations - rho_f, mu_f, rho_s, mu_s, lambda_s, the
    interface load, the inlet amplitude, delta, theta, dt of the solid's d-v row, and three time-level weights (interface load
    lagged towards t^{n-1}, Laplace lifting term on a theta-mix of d^n and d^{n-1}, fluid pressure theta-weighted) - in units of
    the reference's own tolerance.  The response is linear to < 1e-3 of a column, so the joint question is a bounded least-squares
    problem.  What it says:
      * the structural hypotheses (weights = 1 - theta, i.e. a theta-weighted load / pressure) are ruled out by size, and a
        theta-weighted Laplace term is invisible (moves nothing by more than two tolerances and fixes nothing);
      * no one, two or three knobs bring all sixteen numbers inside the reference's tolerances (best triple: 1.10);
      * all thirteen together do (0.45) - with changes of the PROBLEM FILES' constants (rho_s, delta at -2 %, mu_s, lambda_s, the
        load at -0.4 %) that are not free in a restatement, along a direction the data cannot identify: six of the sixteen numbers
        sit inside the tolerance whatever is moved (the cylinder displacements: 2e-5 of 1e-8 against atol 1e-10), which leaves ten
        informative numbers for thirteen unknowns: ANY one of the thirteen can be left out and the other twelve still fit.
    No combination is a restatement of turtleFSI, so nothing is adopted: the strict xfails stay, the study is closed."""
    from itertools import combinations
    from scipy.optimize import lsq_linear
    P, y, A, lin_c, lin_s = _pin_fit()
    assert lin_c < 1e-3 and lin_s < 1e-3
    assert len(P) == 13 and A.shape == (16, 13)
    # where the restatement stands, in the reference's own units: cylinder v_x steps 2, 3 at 3 tolerances, stenosis v_x 2.4, p 70
    assert np.allclose(y[[0, 1, 2, 9, 12]], [-0.63, -2.96, -3.12, 2.44, 70.3], atol=0.06)
    informative = np.abs(y) > 0.02
    assert informative.sum() == 10
    k = {p: i for i, p in enumerate(P)}
    theta1 = 1.0 - THETA
    for name, lo, hi in (("load_lag", 1e3, np.inf), ("pressure_lag", 1e3, np.inf), ("laplace_lag", 0.0, 2.5)):
        moved = np.abs(A[:, k[name]] * theta1).max()
        assert lo <= moved <= hi, (name, moved)                              # theta-weighting: thousands of tolerances, or nothing
    assert np.abs(y + A[:, k["laplace_lag"]] * theta1).max() > 60            # ... and the invisible one fixes nothing
    lb = np.array([-5e-2] * 10 + [-0.5] * 3)                                 # parameters within 5 %, weights within a full theta-mix

    def fit(idx):
        idx = list(idx)
        b = lsq_linear(A[:, idx], -y, bounds=(lb[idx], -lb[idx]))
        return np.abs(y + A[:, idx] @ b.x).max(), b.x

    best = {n: min(fit(c)[0] for c in combinations(range(13), n)) for n in (1, 2, 3)}
    assert best[1] > 3.0 and best[2] > 1.2 and 1.05 < best[3] < 1.2, best     # singles: pressure_lag 3.1; best triple 1.10
    lb2 = np.array([-2e-2] * 10 + [-0.5] * 3)
    full = lsq_linear(A, -y, bounds=(lb2, -lb2))
    r_full = np.abs(y + A @ full.x).max()
    assert r_full < 0.6                                                        # thirteen unknowns, ten informative numbers
    x = dict(zip(P, full.x))
    assert x["rho_s"] < -0.015 and x["delta"] < -0.015 and abs(x["load"]) > 2e-3      # ... by moving what the problem files fix
    loo = {P[i]: lsq_linear(A[:, [j for j in range(13) if j != i]], -y,
                            bounds=(np.delete(lb2, i), -np.delete(lb2, i))) for i in range(13)}
    worst = {p: np.abs(y + np.delete(A, P.index(p), axis=1) @ b.x).max() for p, b in loo.items()}
    assert all(v < 0.7 for v in worst.values()), worst                          # no knob is needed: the data do not identify the fit
