"""The C twin of the oracle's element routines (oracle/fsi_oracle_c.c, OpenMP) against the numpy definition
(oracle/fsi_oracle.py): same inputs, same outputs to round-off, for every material / term the problems use."""
import numpy as np
import pytest

from conftest import GOLDEN, prepare_case


def _states(o, mesh, seed=0):
    rng = np.random.default_rng(seed)
    N2, h = mesh.num_nodes, mesh.hmin()
    U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
    U[:3 * N2] = 0.02 * h * rng.standard_normal(3 * N2)
    U1[:3 * N2] = U[:3 * N2] + 0.002 * h * rng.standard_normal(3 * N2)
    U[3 * N2:6 * N2] = 0.1 * rng.standard_normal(3 * N2)
    U1[3 * N2:6 * N2] = U[3 * N2:6 * N2] + 0.01 * rng.standard_normal(3 * N2)
    U[6 * N2:] = 10 * rng.standard_normal(mesh.num_vertices)
    return U, U1


@pytest.fixture(scope="module")
def c_lib():
    from oracle import c_oracle
    if c_oracle.load() is None:
        pytest.fail("oracle/libfsi_oracle_c.so could not be built (gcc -fopenmp)")
    return c_oracle


def test_c_elements_match_numpy_stvk(cylinder_case, c_lib):
    from oracle.fsi_oracle import FsiOracle
    ns, desc = cylinder_case[0], cylinder_case[1]
    oc, on = FsiOracle(desc), FsiOracle(desc, impl="numpy")
    assert oc.c is not None and on.c is None
    U, U1 = _states(oc, ns["mesh"])
    for a, b in zip(oc.element_residuals(U, U1), on.element_residuals(U, U1)):
        assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()
    sel = np.r_[0:40, len(oc.tets) - 40:len(oc.tets)]          # fluid and solid cells; the numpy Jacobian is the slow side
    d2 = dict(desc)
    for key in ("tets", "tet_nodes", "cell_kind", "cell_region"):
        d2[key] = np.asarray(desc[key])[sel]
    d2.pop("pressure_facets", None); d2.pop("pressure_facet_cell", None); d2.pop("robin_facets", None)
    sc, sn = FsiOracle(d2), FsiOracle(d2, impl="numpy")
    for a, b in zip(sc.element_jacobians(U, U1), sn.element_jacobians(U, U1)):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    Je = sc.element_jacobians(U, U1)[1]
    A, B = sc.assemble_matrix(Je), sn.assemble_matrix(Je)
    assert abs(A - B).max() <= 1e-12 * abs(B).max()
    x = np.random.default_rng(1).standard_normal(A.shape[0])
    assert np.abs(sc.c.spmv(A, x) - A @ x).max() <= 1e-12 * np.abs(A @ x).max()


def test_c_elements_match_numpy_mooney_rivlin(tmp_path, c_lib):
    """avf-type material (two MooneyRivlin regions are exercised by the problem file; here one region with its constants)."""
    from oracle.fsi_oracle import FsiOracle
    ns, desc, *_ = prepare_case("predeform", GOLDEN / "cylinder" / "cylinder.h5", tmp_path)
    assert 1 in list(desc["solid_models"])
    oc, on = FsiOracle(desc), FsiOracle(desc, impl="numpy")
    U, U1 = _states(oc, ns["mesh"], seed=3)
    for a, b in zip(oc.element_residuals(U, U1), on.element_residuals(U, U1)):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    solid = np.nonzero(np.asarray(desc["cell_kind"]) == 1)[0][:24]
    d2 = dict(desc)
    for key in ("tets", "tet_nodes", "cell_kind", "cell_region"):
        d2[key] = np.asarray(desc[key])[solid]
    d2.pop("pressure_facets", None); d2.pop("pressure_facet_cell", None); d2.pop("robin_facets", None)
    sc, sn = FsiOracle(d2), FsiOracle(d2, impl="numpy")
    for a, b in zip(sc.element_jacobians(U, U1), sn.element_jacobians(U, U1)):
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()


def test_full_mesh_kernels_of_the_cpu_baseline_match_the_oracle(tmp_path, c_lib):
    """bench.py's CPU leg on the bench mesh itself (oracle/cpu_port.py: FullMeshKernels - in-place gathers, the sparsity pattern
    from the node graph in C) against the oracle's own routines on the cylinder fixture: the assembled volume residual to
    round-off, the monolithic pattern index for index, the complex-step element Jacobians through their checksums."""
    from oracle.cpu_port import FullMeshKernels, full_mesh_kernels
    from oracle.fsi_oracle import FsiOracle
    ns, desc, *_ = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tmp_path)
    o = FsiOracle(desc)
    k = FullMeshKernels(desc)
    U, U1 = _states(o, ns["mesh"], seed=5)
    Rl, Rn = o.c.residuals(U, U1)
    F_ref = o.c.assemble_vector(Rl + Rn)
    F = k.assemble_residual(U, U1)
    assert np.abs(F - F_ref).max() <= 1e-13 * np.abs(F_ref).max()
    indptr, indices, data = k.pattern()
    rp, ri = o.c.pattern()
    assert np.array_equal(indptr, rp) and np.array_equal(indices, ri) and np.all(data > 0.0)
    Jl, Jn = o.c.jacobians(U, U1)
    cs = k.jacobian_elements(U, U1, 200, 232)
    ref = (Jl + Jn)[200:232].sum(axis=(1, 2))
    assert np.abs(cs - ref).max() <= 1e-10 * np.abs(ref).max()
    x = np.random.default_rng(0).standard_normal(k.ndof)
    import scipy.sparse as sp
    A = sp.csr_matrix((data, indices, indptr), shape=(k.ndof, k.ndof))
    assert np.abs(k.spmv(indptr, indices, data, x, np.empty(k.ndof)) - A @ x).max() <= 1e-12 * np.abs(A @ x).max()
    r = full_mesh_kernels(desc, budget_s=0.6)
    assert r["tets"] == 1647 and r["spmv"]["nnz"] == len(indices) and r["residual_assembly"]["tets_per_s"] > 0
    assert r["jacobian_elements"]["tets"] > 0
