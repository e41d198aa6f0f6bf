"""Tet-mesh topology and the mixed P2/P2/P1 dof layout of the monolithic (d, v, p) space.

Host-side (numpy, one-off) counterpart of what DOLFIN builds when the reference driver does
``DVP = FunctionSpace(mesh, MixedElement([P2^3, P2^3, P1]))`` [SURVEY.md §3.1, §8a row a3] after
``get_mesh_domain_and_boundaries`` has read ``/mesh``, ``/boundaries`` and ``/domains``
[REF src/vasp/simulations/offset_stenosis.py:85-94].

Conventions (shared by the HIP kernels, the C-ABI and the oracle; stated once here):

* cells are vertex-sorted rows (DOLFIN convention), so the affine map may have
  either orientation;
* P2 nodes: ``0..V-1`` are the mesh vertices, ``V..V+E-1`` the edge midpoints, edges numbered in
  lexicographic order of their sorted vertex pair;
* local P2 order on a tet: 4 vertices, then edges in UFC order
  e0=(v2,v3) e1=(v1,v3) e2=(v1,v2) e3=(v0,v3) e4=(v0,v2) e5=(v0,v1);
* local P2 order on a facet (a<b<c): a, b, c, then edges (b,c), (a,c), (a,b);
* global dof layout (field-major, pressure last so incomplete factorisations meet the saddle-point
  zeros after the velocity pivots): ``[ d: 3*N2 | v: 3*N2 | p: V ]`` with ``d[3*node + comp]``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Optional

import numpy as np

from .h5lite import Dataset, Group, read_h5, write_h5

TET_EDGES = np.array([[2, 3], [1, 3], [1, 2], [0, 3], [0, 2], [0, 1]], dtype=np.int64)
TRI_EDGES = np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int64)
TET_FACES = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=np.int64)  # face i is opposite vertex i

FIELD_D, FIELD_V, FIELD_P = 0, 1, 2


def _pair_keys(a: np.ndarray, b: np.ndarray, n: int) -> np.ndarray:
    lo = np.minimum(a, b).astype(np.int64)
    hi = np.maximum(a, b).astype(np.int64)
    return lo * n + hi


@dataclass
class FsiMesh:
    coords: np.ndarray          # (V,3) f64
    tets: np.ndarray            # (C,4) i64, rows ascending
    cell_markers: np.ndarray    # (C,) i64   (``domains``)
    facets: np.ndarray          # (F,3) i64, rows ascending
    facet_markers: np.ndarray   # (F,) i64   (``boundaries``)
    # derived ------------------------------------------------------------------------------
    edges: np.ndarray = field(default=None, repr=False)        # (E,2)
    tet_nodes: np.ndarray = field(default=None, repr=False)    # (C,10) P2 node ids
    facet_nodes: np.ndarray = field(default=None, repr=False)  # (F,6)  P2 node ids
    facet_cells: np.ndarray = field(default=None, repr=False)  # (F,2)  incident cells, -1 = none
    node_coords: np.ndarray = field(default=None, repr=False)  # (N2,3)

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, coords, tets, cell_markers, facets=None, facet_markers=None) -> "FsiMesh":
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        tets = np.sort(np.asarray(tets, dtype=np.int64), axis=1)
        cell_markers = np.asarray(cell_markers, dtype=np.int64).copy()
        if facets is None:
            allf = np.sort(tets[:, TET_FACES].reshape(-1, 3), axis=1)
            facets = np.unique(allf, axis=0)
            facet_markers = np.zeros(len(facets), dtype=np.int64)
        facets = np.sort(np.asarray(facets, dtype=np.int64), axis=1)
        facet_markers = np.asarray(facet_markers).astype(np.int64).copy()
        m = cls(coords, tets, cell_markers, facets, facet_markers)
        m._build()
        return m

    @classmethod
    def read(cls, path) -> "FsiMesh":
        """Read the DOLFIN HDF5 mesh layout (``/mesh``, ``/boundaries``, ``/domains``)."""
        g = read_h5(path)
        coords = g["mesh"]["coordinates"].data
        tets = g["mesh"]["topology"].data
        dom_topo = np.sort(g["domains"]["topology"].data, axis=1)
        dom_vals = g["domains"]["values"].data.astype(np.int64)
        tets_sorted = np.sort(tets, axis=1)
        if dom_topo.shape == tets_sorted.shape and np.array_equal(dom_topo, tets_sorted):
            cell_markers = dom_vals
        else:  # match cells by their vertex tuple, as DOLFIN does when reading a MeshFunction
            V = len(coords)
            key = lambda t: ((t[:, 0] * V + t[:, 1]) * V + t[:, 2]) * V + t[:, 3]
            order = np.argsort(key(dom_topo))
            pos = np.searchsorted(key(dom_topo)[order], key(tets_sorted))
            cell_markers = dom_vals[order[pos]]
        facets = g["boundaries"]["topology"].data
        fvals = g["boundaries"]["values"].data
        # DOLFIN stores "unset" size_t markers as 2^64-1; keep them as -1
        fm = fvals.astype(np.int64)
        return cls.from_arrays(coords, tets, cell_markers, facets, fm)

    def _build(self):
        V = len(self.coords)
        t = self.tets
        # edges -----------------------------------------------------------------------------
        ea = t[:, TET_EDGES[:, 0]]
        eb = t[:, TET_EDGES[:, 1]]
        keys = _pair_keys(ea, eb, V)
        ukeys, inv = np.unique(keys.ravel(), return_inverse=True)
        self.edges = np.stack([ukeys // V, ukeys % V], axis=1)
        tet_edges = inv.reshape(keys.shape)
        self.tet_nodes = np.concatenate([t, V + tet_edges], axis=1)
        self.node_coords = np.concatenate(
            [self.coords, 0.5 * (self.coords[self.edges[:, 0]] + self.coords[self.edges[:, 1]])], axis=0)
        # facets ----------------------------------------------------------------------------
        f = self.facets
        fk = _pair_keys(f[:, TRI_EDGES[:, 0]], f[:, TRI_EDGES[:, 1]], V)
        pos = np.searchsorted(ukeys, fk)
        if not np.array_equal(ukeys[np.minimum(pos, len(ukeys) - 1)], fk):
            raise ValueError("facet list references an edge that is not in any cell")
        self.facet_nodes = np.concatenate([f, V + pos], axis=1)
        # facet -> cells
        cf = np.sort(t[:, TET_FACES], axis=2)                       # (C,4,3)
        ckey = (cf[:, :, 0] * V + cf[:, :, 1]) * V + cf[:, :, 2]      # (C,4)
        fkey = (f[:, 0] * V + f[:, 1]) * V + f[:, 2]
        forder = np.argsort(fkey)
        loc = np.searchsorted(fkey[forder], ckey.ravel())
        loc = np.minimum(loc, len(fkey) - 1)
        ok = fkey[forder][loc] == ckey.ravel()
        fid = forder[loc]
        cell_of = np.repeat(np.arange(len(t)), 4)
        self.facet_cells = -np.ones((len(f), 2), dtype=np.int64)
        order = np.argsort(fid[ok], kind="stable")
        fids, cells = fid[ok][order], cell_of[ok][order]
        first = np.r_[True, fids[1:] != fids[:-1]]
        self.facet_cells[fids[first], 0] = cells[first]
        second = ~first
        self.facet_cells[fids[second], 1] = cells[second]

    # ---- sizes / dof layout -------------------------------------------------------------------
    @property
    def num_vertices(self) -> int:
        return len(self.coords)

    @property
    def num_cells(self) -> int:
        return len(self.tets)

    @property
    def num_nodes(self) -> int:
        """Number of P2 nodes (V + E)."""
        return len(self.node_coords)

    @property
    def num_dofs(self) -> int:
        return 6 * self.num_nodes + self.num_vertices

    def field_offset(self, fld: int) -> int:
        return (0, 3 * self.num_nodes, 6 * self.num_nodes)[fld]

    def dofs(self, fld: int, nodes: np.ndarray, comp: Optional[int] = None) -> np.ndarray:
        """Global dofs of ``fld`` at P2 ``nodes`` (vertices for p). ``comp`` None → all 3 (node-major)."""
        nodes = np.asarray(nodes, dtype=np.int64)
        if fld == FIELD_P:
            return self.field_offset(FIELD_P) + nodes
        base = self.field_offset(fld) + 3 * nodes
        if comp is None:
            return (base[:, None] + np.arange(3)).ravel()
        return base + comp

    def cell_dofs(self) -> np.ndarray:
        """(C,64) global dofs in the local order [d_x(10) d_y d_z v_x v_y v_z p(4)]."""
        tn = self.tet_nodes
        N2 = self.num_nodes
        cols = []
        for fld in (FIELD_D, FIELD_V):
            for c in range(3):
                cols.append(self.field_offset(fld) + 3 * tn + c)
        cols.append(6 * N2 + self.tets)
        return np.concatenate(cols, axis=1)

    def split(self, x: np.ndarray):
        """Views (d (N2,3), v (N2,3), p (V,)) of a global vector."""
        N2 = self.num_nodes
        return x[:3 * N2].reshape(N2, 3), x[3 * N2:6 * N2].reshape(N2, 3), x[6 * N2:]

    # ---- geometry helpers ----------------------------------------------------------------------
    def cell_midpoints(self) -> np.ndarray:
        return self.coords[self.tets].mean(axis=1)

    def facet_midpoints(self) -> np.ndarray:
        return self.coords[self.facets].mean(axis=1)

    def facet_area_normals(self, fids: np.ndarray, cell_side: Optional[np.ndarray] = None):
        """Area and unit normal of ``fids``; the normal points out of ``cell_side`` (default: cell 0)."""
        fids = np.asarray(fids, dtype=np.int64)
        f = self.facets[fids]
        x = self.coords
        nvec = np.cross(x[f[:, 1]] - x[f[:, 0]], x[f[:, 2]] - x[f[:, 0]])
        area = 0.5 * np.linalg.norm(nvec, axis=1)
        n = nvec / (2.0 * area)[:, None]
        cells = self.facet_cells[fids, 0] if cell_side is None else np.asarray(cell_side)
        ccen = x[self.tets[cells]].mean(axis=1)
        fcen = x[f].mean(axis=1)
        flip = np.einsum("ij,ij->i", n, fcen - ccen) < 0
        n[flip] *= -1.0
        return area, n

    def hmin(self) -> float:
        """DOLFIN ``mesh.hmin()``: min over cells of the circumradius*2 (cell diameter).  The mesh never moves (ALE on the
        reference configuration), so the value is computed once."""
        cached = getattr(self, "_hmin", None)
        if cached is not None:
            return cached
        x = self.coords[self.tets]
        a = x[:, 1] - x[:, 0]
        b = x[:, 2] - x[:, 0]
        c = x[:, 3] - x[:, 0]
        num = (np.einsum("ij,ij->i", a, a)[:, None] * np.cross(b, c)
               + np.einsum("ij,ij->i", b, b)[:, None] * np.cross(c, a)
               + np.einsum("ij,ij->i", c, c)[:, None] * np.cross(a, b))
        den = 2.0 * np.einsum("ij,ij->i", a, np.cross(b, c))
        r = np.linalg.norm(num, axis=1) / np.abs(den)
        self._hmin = float(2.0 * r.min())
        return self._hmin

    # ---- point location (probes) -----------------------------------------------------------------
    def locate(self, pts: np.ndarray, tol: float = 1e-12):
        """First cell containing each point → (cell or -1, barycentric (n,4)).

        Mirrors DOLFIN point evaluation with ``allow_extrapolation=False``: a point in no cell
        raises there [REF src/vasp/simulations/simulation_common.py:213-216]; here it is cell -1.
        """
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        x = self.coords[self.tets]
        T = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2)  # (C,3,3)
        Tinv = np.linalg.inv(T)
        cells = -np.ones(len(pts), dtype=np.int64)
        bary = np.zeros((len(pts), 4))
        lo = x.min(axis=1)
        hi = x.max(axis=1)
        for i, p in enumerate(pts):
            cand = np.nonzero(np.all((p >= lo - 1e-14) & (p <= hi + 1e-14), axis=1))[0]
            if len(cand) == 0:
                continue
            l = np.einsum("cij,cj->ci", Tinv[cand], p - x[cand, 0])
            lam = np.concatenate([1.0 - l.sum(axis=1, keepdims=True), l], axis=1)
            inside = np.nonzero(np.all(lam >= -tol, axis=1))[0]
            if len(inside):
                cells[i] = cand[inside[0]]
                bary[i] = lam[inside[0]]
        return cells, bary

    # ---- output ------------------------------------------------------------------------------------
    def to_h5_group(self) -> Group:
        """DOLFIN ``HDF5File`` layout with the *current* (relabelled) markers."""
        root = Group()
        V = self.coords
        topo_attrs = lambda ct: {"celltype": ct, "partition": np.array([0], dtype=np.uint64)}
        C = len(self.tets)
        m = Group()
        m["coordinates"] = Dataset(V.copy())
        m["topology"] = Dataset(self.tets.astype(np.int64), topo_attrs("tetrahedron"))
        m["cell_indices"] = Dataset(np.arange(C, dtype=np.int64))
        b = Group()
        b["coordinates"] = Dataset(V.copy())
        b["topology"] = Dataset(self.facets.astype(np.int64), topo_attrs("triangle"))
        b["values"] = Dataset(self.facet_markers.astype(np.uint64))
        d = Group()
        d["coordinates"] = Dataset(V.copy())
        d["topology"] = Dataset(self.tets.astype(np.int64), topo_attrs("tetrahedron"))
        d["cell_indices"] = Dataset(np.arange(C, dtype=np.int64))
        d["values"] = Dataset(self.cell_markers.astype(np.uint64))
        root["mesh"], root["boundaries"], root["domains"] = m, b, d
        return root

    def write(self, path) -> None:
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        write_h5(path, self.to_h5_group())


# ------------------------------------------------------------------------------------------------
# P2 / P1 nodal interpolation at a located point
# ------------------------------------------------------------------------------------------------

def p2_basis(lam: np.ndarray) -> np.ndarray:
    """P2 basis values (…,10) at barycentric coordinates (…,4) in the local order above."""
    l = lam
    vert = l * (2.0 * l - 1.0)
    edge = 4.0 * l[..., TET_EDGES[:, 0]] * l[..., TET_EDGES[:, 1]]
    return np.concatenate([vert, edge], axis=-1)
