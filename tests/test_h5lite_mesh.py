"""Input-mesh format and topology: the libhdf5-free reader/writer against the reference's own mesh fixtures
(copied data files: tests/test_data/{offset_stenosis,cylinder,aneurysm} of the reference) and the P2/P1 layout.

Sizes pinned by SURVEY.md §4 (read off the fixtures) and REF tests/test_postprocess_mesh.py:33-42 (2 500 = V + E of the
cylinder)."""
import numpy as np
import pytest

from conftest import GOLDEN
from vasp_amd.h5lite import read_h5, write_h5
from vasp_amd.mesh import FsiMesh, p2_basis

FIXTURES = {
    "offset_stenosis/offset_stenosis.h5": dict(V=1287, C=6590, F=13571, N2=9554, ndof=58611, fluid=4616, solid=1974),
    "cylinder/cylinder.h5": dict(V=352, C=1647, F=3444, N2=2500, ndof=15352, fluid=1128, solid=519),
    "aneurysm/small_aneurysm.h5": dict(V=1224, C=6073, F=12582, N2=8956, ndof=54960, fluid=3979, solid=2094),
}


@pytest.mark.parametrize("name", list(FIXTURES))
def test_read_fixture_sizes(name):
    exp = FIXTURES[name]
    m = FsiMesh.read(GOLDEN / name)
    assert (m.num_vertices, m.num_cells, len(m.facets)) == (exp["V"], exp["C"], exp["F"])
    assert m.num_nodes == exp["N2"] and m.num_dofs == exp["ndof"]
    assert (m.cell_markers == 1).sum() == exp["fluid"] and (m.cell_markers == 2).sum() == exp["solid"]
    assert np.all(np.diff(m.tets, axis=1) > 0) and np.all(np.diff(m.facets, axis=1) > 0)     # DOLFIN: rows ascending
    # every facet has 1 or 2 incident cells; interior ones exactly 2
    assert np.all(m.facet_cells[:, 0] >= 0)
    # Euler characteristic of a ball-like tet mesh with E from the P2 count: V - E + F - C = 1
    E = m.num_nodes - m.num_vertices
    assert m.num_vertices - E + len(m.facets) - m.num_cells == 1


def test_dataset_names_dtypes_attrs():
    g = read_h5(GOLDEN / "cylinder" / "cylinder.h5")
    assert sorted(g) == ["boundaries", "domains", "mesh"]
    assert sorted(g["mesh"]) == ["cell_indices", "coordinates", "topology"]
    assert sorted(g["boundaries"]) == ["coordinates", "topology", "values"]
    assert sorted(g["domains"]) == ["cell_indices", "coordinates", "topology", "values"]
    assert g["mesh"]["coordinates"].data.dtype == np.float64
    assert g["mesh"]["topology"].data.dtype == np.int64
    assert g["boundaries"]["values"].data.dtype == np.uint64
    ct = g["mesh"]["topology"].attrs["celltype"]
    assert (ct.decode() if isinstance(ct, bytes) else str(ct)).rstrip("\0") == "tetrahedron"
    assert np.asarray(g["mesh"]["topology"].attrs["partition"]).tolist() == [0]


def test_write_read_round_trip(tmp_path):
    src = GOLDEN / "cylinder" / "cylinder.h5"
    m = FsiMesh.read(src)
    m.write(tmp_path / "mesh.h5")
    a, b = read_h5(src), read_h5(tmp_path / "mesh.h5")
    for grp in ("mesh", "boundaries", "domains"):
        for name in a[grp]:
            assert np.array_equal(np.asarray(a[grp][name].data), np.asarray(b[grp][name].data)), (grp, name)
            assert a[grp][name].data.dtype == b[grp][name].data.dtype
    raw = (tmp_path / "mesh.h5").read_bytes()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0           # superblock version 0, as DOLFIN's files


def test_relabel_rules_offset_stenosis(stenosis_case):
    """get_mesh_domain_and_boundaries [REF offset_stenosis.py:98-138] on the fixture: 743 wall facets -> rigid,
    77 fluid cells -> 1001."""
    ns = stenosis_case[0]
    b, d = ns["boundaries"], ns["domains"]
    assert (b == 22).sum() == 290 and (b == 33).sum() == 283 and (b == 11).sum() == 48 + 743
    assert (d == 1001).sum() == 77 and (d == 1).sum() == 4616 - 77


def test_p2_layout_and_location():
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    # edge nodes sit at edge midpoints, UFC order e0=(v2,v3) ... e5=(v0,v1)
    tn = m.tet_nodes
    x = m.node_coords
    for e, (i, j) in enumerate([(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]):
        assert np.allclose(x[tn[:, 4 + e]], 0.5 * (x[tn[:, i]] + x[tn[:, j]]))
    # P2 interpolation reproduces a quadratic exactly at located points
    f = lambda p: 1 + 2 * p[..., 0] - p[..., 1] * p[..., 2] + 3 * p[..., 0] ** 2
    pts = m.cell_midpoints()[::97] + 1e-5
    cells, lam = m.locate(pts)
    assert np.all(cells >= 0)
    vals = np.einsum("na,na->n", p2_basis(lam), f(x[tn[cells]]))
    assert np.allclose(vals, f(pts), rtol=1e-12, atol=1e-14)
    assert m.locate(np.array([[1.0, 1.0, 1.0]]))[0][0] == -1          # outside: no cell


def test_generated_mesh_is_conforming_and_marked(tmp_path):
    from vasp_amd.meshgen import write_mesh
    write_mesh(tmp_path / "s.h5", 6000)
    m = FsiMesh.read(tmp_path / "s.h5")
    x = m.coords[m.tets]
    vol = np.abs(np.linalg.det(np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2))) / 6
    assert vol.min() > 1e-3 * vol.max()
    interior = m.facet_cells[:, 1] >= 0
    assert set(np.unique(m.facet_markers[~interior])) == {2, 3, 11, 33}
    assert set(np.unique(m.facet_markers[interior])) == {0, 22}
    mk = m.cell_markers[m.facet_cells[m.facet_markers == 22]]
    assert np.all(np.sort(mk, axis=1) == [1, 2])                       # 22 separates fluid from solid
    assert m.num_vertices - (m.num_nodes - m.num_vertices) + len(m.facets) - m.num_cells == 1
