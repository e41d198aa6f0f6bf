"""Self-checks of the CPU oracle's building blocks (quadrature, tabulation, element residuals and their tangents).

These are properties the restated formulas must have whatever the reference's numbers are; the pins against the
reference's own known answers are in test_oracle_pins.py."""
import numpy as np
import pytest

from oracle.fsi_oracle import FsiOracle, keast24, tabulate_p2, triangle12


def test_keast24_is_exact_to_degree_6():
    from math import factorial
    pts, w = keast24()
    assert pts.shape == (24, 3) and np.isclose(w.sum(), 1 / 6)
    for a in range(7):
        for b in range(7 - a):
            for c in range(7 - a - b):
                exact = factorial(a) * factorial(b) * factorial(c) / factorial(a + b + c + 3)
                assert np.isclose((w * pts[:, 0] ** a * pts[:, 1] ** b * pts[:, 2] ** c).sum(), exact, rtol=1e-12), (a, b, c)


def test_triangle12_is_exact_to_degree_6():
    from math import factorial
    pts, w = triangle12()
    assert np.isclose(w.sum(), 0.5)
    for a in range(7):
        for b in range(7 - a):
            exact = factorial(a) * factorial(b) / factorial(a + b + 2)
            assert np.isclose((w * pts[:, 0] ** a * pts[:, 1] ** b).sum(), exact, rtol=1e-10), (a, b)


def test_p2_tabulation():
    pts, _ = keast24()
    N, dN, L, dL = tabulate_p2(pts)
    assert np.allclose(N.sum(axis=1), 1) and np.allclose(dN.sum(axis=1), 0)
    # nodal property at the 10 P2 nodes of the reference tet (UFC order)
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    edges = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
    nodes = np.concatenate([v, [0.5 * (v[i] + v[j]) for i, j in edges]])
    assert np.allclose(tabulate_p2(nodes)[0], np.eye(10))


def _two_tet_desc(dt=1e-3, theta=0.51):
    """One fluid and one solid tet sharing a face, the second with negative orientation (DOLFIN sorts vertices)."""
    from vasp_amd.mesh import FsiMesh
    coords = np.array([[0, 0, 0], [1e-3, 0, 0], [0, 1.2e-3, 0], [1e-4, 2e-4, 0.9e-3], [3e-4, 2e-4, -1.1e-3]])
    m = FsiMesh.from_arrays(coords, [[0, 1, 2, 3], [0, 1, 2, 4]], [1, 2])
    kind = np.array([0, 1])
    return m, dict(coords=m.coords, tets=m.tets, tet_nodes=m.tet_nodes, num_nodes=m.num_nodes, cell_kind=kind,
                   cell_region=np.zeros(2, int), fluid_props=[(1000.0, 1.5e-3)], solid_props=[(1000.0, 344827.6, 3103448.3)],
                   dt=dt, theta=theta)


def _random_state(o, m, seed=0):
    rng = np.random.default_rng(seed)
    N2, h = m.num_nodes, 1e-3
    U = np.zeros(o.ndof)
    U[:3 * N2] = 0.03 * h * rng.standard_normal(3 * N2)
    U[3 * N2:6 * N2] = 0.1 * rng.standard_normal(3 * N2)
    U[6 * N2:] = 10 * rng.standard_normal(m.num_vertices)
    U1 = U.copy()
    U1[:6 * N2] += 0.1 * U[:6 * N2] * rng.standard_normal(6 * N2)
    return U, U1


def test_orientations_and_zero_state():
    m, desc = _two_tet_desc()
    o = FsiOracle(desc)
    assert o.detJ[0] * o.detJ[1] < 0                       # both orientations are exercised
    assert np.allclose(o.wdet.sum(axis=1), np.abs(o.detJ) / 6)
    z = np.zeros(o.ndof)
    assert np.abs(o.residual(z, z)).max() == 0.0           # F(0; 0) = 0 without loads


def test_complex_step_jacobian_matches_finite_differences():
    m, desc = _two_tet_desc()
    o = FsiOracle(desc)
    U, U1 = _random_state(o, m)
    Jl, Jn = o.element_jacobians(U, U1)
    loc, loc1 = o.gather(U), o.gather(U1)
    scale = np.abs(Jl + Jn).max(axis=(1, 2))
    for j in range(0, 64, 5):
        e = np.zeros(64)
        e[j] = 1
        step = 1e-6 * max(1.0, abs(loc[:, j]).max())
        for c, fn in enumerate([lambda a, b: o._fluid_residual(np.array([0]), 1000.0, 1.5e-3, a, b),
                                lambda a, b: o._solid_residual(np.array([1]), 1000.0, 344827.6, 3103448.3, a, b)]):
            rp = sum(fn(loc[c:c + 1] + step * e, loc1[c:c + 1]))
            rm = sum(fn(loc[c:c + 1] - step * e, loc1[c:c + 1]))
            fd = (rp - rm)[0] / (2 * step)
            assert np.abs(fd - (Jl + Jn)[c, :, j]).max() < 2e-5 * scale[c], (c, j)


def test_linear_part_is_linear_in_the_new_state():
    """F_linear must be affine in dvp_n (turtleFSI assembles its Jacobian once, as A_pre)."""
    m, desc = _two_tet_desc()
    o = FsiOracle(desc)
    U, U1 = _random_state(o, m)
    Ua, _ = _random_state(o, m, seed=1)
    Jl_a, _ = o.element_jacobians(U, U1)
    Jl_b, _ = o.element_jacobians(Ua, U1)
    assert np.abs(Jl_a - Jl_b).max() <= 1e-9 * np.abs(Jl_a).max()


def test_theta_scheme_identity_in_the_solid():
    """The delta-penalised d-equation is (d_n - d_{n-1})/k = theta v_n + (1-theta) v_{n-1} tested with phi:
    a state that satisfies it nodally has a vanishing d-residual in the solid cell (SURVEY.md A.1 check)."""
    m, desc = _two_tet_desc()
    o = FsiOracle(desc)
    U, U1 = _random_state(o, m)
    N2, k, th = m.num_nodes, desc["dt"], desc["theta"]
    U[:3 * N2] = U1[:3 * N2] + k * (th * U[3 * N2:6 * N2] + (1 - th) * U1[3 * N2:6 * N2])
    Rl, Rn = o.element_residuals(U, U1)
    dres = (Rl + Rn)[1, :30]
    ref = 1e7 * 1000.0 * np.abs(U[3 * N2:6 * N2]).max() * np.abs(o.detJ[1]) / 6
    assert np.abs(dres).max() < 1e-10 * ref


def test_pressure_load_integrates_to_P_times_area_normal():
    from conftest import GOLDEN, prepare_case
    import tempfile
    ns, desc, bc_values, pressure, hook = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tempfile.mkdtemp())
    o = FsiOracle(desc)
    z = np.zeros(o.ndof)
    R = o.facet_residual(z, z, 2.0)
    N2 = o.N2
    total = R[3 * N2:6 * N2].reshape(N2, 3).sum(axis=0)
    assert np.allclose(total, 2.0 * o.pf_nA.sum(axis=0), rtol=1e-12)
    # the '+' side is the solid cell: its outward normal points into the lumen, i.e. towards the axis on average
    mesh = ns["mesh"]
    fc = mesh.coords[desc["pressure_facets"][:, :3]].mean(axis=1)
    axis_pt = mesh.coords.mean(axis=0)
    radial = fc - axis_pt
    # project out the axial direction (largest extent)
    ax = np.argmax(np.ptp(mesh.coords, axis=0))
    radial[:, ax] = 0
    assert (np.einsum("ij,ij->i", o.pf_nA, radial) < 0).mean() > 0.95


def test_mooney_rivlin_reduces_to_hooke_for_small_strain():
    """psi = C10 (I1b-3) + C01 (I2b-3) + ... + K (J ln J - J + 1): for infinitesimal strain the stress is that of an
    isotropic solid with shear modulus 2 (C10 + C01) and bulk modulus K = lambda + 2 mu / 3 (C11 enters at higher order)."""
    m, desc = _two_tet_desc()
    mu, lam, C10, C01 = 344827.6, 3103448.3, 1.0e5, 0.7e5
    desc_mr = dict(desc, solid_props=[(1000.0, mu, lam, C10, C01, 3.0e6)], solid_models=[1])
    G = 2 * (C10 + C01)
    K = lam + 2 * mu / 3
    desc_svk = dict(desc, solid_props=[(1000.0, G, K - 2 * G / 3)], solid_models=[0])
    o_mr, o_svk = FsiOracle(desc_mr), FsiOracle(desc_svk)
    rng = np.random.default_rng(5)
    U = np.zeros(o_mr.ndof)
    U[:3 * m.num_nodes] = 1e-9 * rng.standard_normal(3 * m.num_nodes)      # strain ~ 1e-6
    z = np.zeros_like(U)
    r_mr = o_mr.element_residuals(U, z)[1][1]                               # nonlinear part of the solid cell
    r_svk = o_svk.element_residuals(U, z)[1][1]
    assert np.abs(r_mr - r_svk).max() < 1e-4 * np.abs(r_svk).max()
    # and its tangent (complex step) is symmetric in the v-test / d-trial block, as a hyperelastic tangent must be
    Jn = o_mr.element_jacobians(U, z)[1][1]
    Kvd = Jn[30:60, 0:30]
    assert np.abs(Kvd - Kvd.T).max() < 1e-8 * np.abs(Kvd).max()
