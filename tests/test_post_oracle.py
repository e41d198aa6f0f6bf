"""The post-processing oracle (oracle/post_oracle.py: stress / strain / WSS of SURVEY.md §8f-4) on closed-form cases."""
import numpy as np

from conftest import GOLDEN


def fluid_boundary_facets(mesh, fluid_markers=(1,)):
    """Exterior facets of the fluid sub-mesh: (cell, local index of the opposite vertex) per facet."""
    fc = mesh.facet_cells
    is_f = np.isin(mesh.cell_markers, fluid_markers)
    c0, c1 = fc[:, 0], fc[:, 1]
    f0 = is_f[c0]
    f1 = np.where(c1 >= 0, is_f[np.maximum(c1, 0)], False)
    sel = np.nonzero(f0 ^ f1)[0]                       # exactly one fluid cell: boundary of the fluid sub-mesh
    cell = np.where(f0[sel], c0[sel], c1[sel])
    tv = mesh.tets[cell]
    local = np.array([int(np.nonzero(~np.isin(tv[i], mesh.facets[sel[i]]))[0][0]) for i in range(len(sel))])
    return sel, cell, local


def test_kopp_root_is_the_largest_eigenvalue():
    from oracle.post_oracle import kopp_max_eigenvalue
    rng = np.random.default_rng(0)
    A = rng.standard_normal((200, 3, 3)) * 10.0 ** rng.integers(-6, 6, size=(200, 1, 1))
    T = A + np.swapaxes(A, -1, -2)
    ref = np.linalg.eigvalsh(T)[:, -1]
    assert np.abs(kopp_max_eigenvalue(T) - ref).max() <= 1e-9 * np.abs(ref).max()
    rel = np.abs(kopp_max_eigenvalue(T) - ref) / np.abs(T).max(axis=(1, 2))
    assert rel.max() < 1e-10


def test_homogeneous_deformation_gives_constant_dg1_tensors():
    from oracle.post_oracle import stress_strain_dg1
    from vasp_amd.mesh import FsiMesh
    mesh = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    A = np.array([[0.02, 0.01, 0.0], [-0.005, 0.03, 0.004], [0.0, 0.002, -0.01]])
    d = mesh.node_coords @ A.T
    solid = np.nonzero(mesh.cell_markers == 2)[0][:50]
    mu, lam = 344827.586, 3103448.27
    out = stress_strain_dg1(mesh.coords, mesh.tets, mesh.tet_nodes, d, solid, (1e3, mu, lam))
    F = np.eye(3) + A
    E = 0.5 * (F.T @ F - np.eye(3))
    S = lam * np.trace(E) * np.eye(3) + 2 * mu * E
    sigma = F @ S @ F.T / np.linalg.det(F)
    assert np.abs(out["GreenLagrangeStrain"] - E).max() < 1e-12
    assert np.abs(out["TrueStress"] - sigma).max() < 1e-9 * np.abs(sigma).max()
    assert np.abs(out["MaxPrincipalStress"] - np.linalg.eigvalsh(0.5 * (sigma + sigma.T))[-1]).max() < 1e-9 * np.abs(sigma).max()
    assert np.abs(out["MaxPrincipalStrain"] - np.linalg.eigvalsh(E)[-1]).max() < 1e-12


def test_linear_velocity_gives_the_analytic_tangential_traction():
    from oracle.post_oracle import wall_shear_stress
    from vasp_amd.mesh import FsiMesh
    mesh = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    B = np.array([[0.0, 2.0, 1.0], [0.5, 0.0, -1.0], [0.3, 0.7, 0.0]])
    v = mesh.node_coords @ B.T
    fids, cell, local = fluid_boundary_facets(mesh)
    assert len(fids) == 173 + 36 + 41                   # interface + inlet + outlet of the cylinder fixture
    mu = 3.5e-3
    out = wall_shear_stress(mesh.coords, mesh.tets, mesh.tet_nodes, v, cell, local, mu)
    single = np.array([np.sum(cell == c) == 1 for c in cell])
    assert single.sum() > 100
    area, n = mesh.facet_area_normals(fids)
    centre = mesh.coords[mesh.tets[cell]].mean(axis=1)
    flip = np.einsum("fi,fi->f", n, mesh.coords[mesh.facets[fids]].mean(axis=1) - centre) < 0
    n = np.where(flip[:, None], -n, n)
    Fv = -np.einsum("ij,fj->fi", mu * (B + B.T), n)
    Ft = Fv - np.einsum("fi,fi->f", Fv, n)[:, None] * n
    assert np.abs(out[single] - Ft[single][:, None, :]).max() < 1e-12 * np.abs(Ft).max()
