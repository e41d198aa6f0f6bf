"""Deterministic generator of offset-stenosis FSI meshes at any size.

The reference tree holds only 1.6 k - 12 k-tet fixtures; the tutorial's larger meshes are downloads
[REF docs/offset_stenosis.md:122-127].  BASELINE.json's configurations (~50 k, 1 M, 3 M, 10 M tets) are therefore
produced here, in the file layout ``vasp-generate-mesh`` emits (``/mesh``, ``/boundaries``, ``/domains``; fluid = 1,
solid = 2 [REF src/vasp/preprocessing/preprocessing_common.py:198-247]) and with the boundary ids the offset-stenosis
problem file expects (inlet 3, outlet 2, solid ends 11, interface 22, outer wall 33
[REF src/vasp/simulations/offset_stenosis.py:51-56]); the problem file's own relabelling rules (:98-138) then apply
unchanged.  Geometry: the eccentric cosine stenosis of Varghese et al. (2007) that the tutorial cites
[REF docs/offset_stenosis.md:5] scaled to the fixture's bounding box (x in [-9.5, 24.9] mm, lumen radius 3.45 mm)
plus one solid layer 0.3 mm thick.

Construction: an O-grid cross-section (central square block + ring layers that blend from square to circle; the last
ring is the vessel wall), extruded along x; every hexahedron is cut into 12 tetrahedra around its centre, each face
along the diagonal through its lowest-numbered vertex, which makes neighbouring cells conform without any lookup.
Interior vertices are jittered by a seeded pseudo-random offset so that the mesh is not a lattice.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from .h5lite import Dataset, Group, write_h5

X_MIN, X_MAX = -0.0095, 0.0249
R_LUMEN, T_WALL = 0.00345, 0.0003
HEX_FACES = np.array([[0, 1, 3, 2], [4, 5, 7, 6], [0, 1, 5, 4], [2, 3, 7, 6], [0, 2, 6, 4], [1, 3, 7, 5]])


def _stenosis(x):
    """Lumen radius scale and axis offset (y) at axial positions x: 75 % area reduction, 5 % D eccentricity."""
    D = 2 * R_LUMEN
    s = np.where(np.abs(x) < D, 0.25 * (1 + np.cos(np.pi * x / D)), 0.0)     # length 2D centred at x = 0
    return 1.0 - s, 0.1 * s * D


def _cross_section(nc: int, nr: int):
    """O-grid on the unit disc: points (n,2), quads (m,4) [p00,p10,p01,p11], ring index per quad (-1 = core)."""
    half = 0.4                                             # core square half-width relative to the unit radius
    g = np.linspace(-half, half, nc + 1)
    core = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    cid = lambda i, j: i * (nc + 1) + j
    # loop of 4*nc nodes around the core boundary, counter-clockwise from (-half,-half)
    loop0 = ([cid(i, 0) for i in range(nc)] + [cid(nc, j) for j in range(nc)]
             + [cid(nc - i, nc) for i in range(nc)] + [cid(0, nc - j) for j in range(nc)])
    sq = core[loop0]
    ang = np.arctan2(sq[:, 1], sq[:, 0])
    circ = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    pts = [core]
    loops = [np.array(loop0)]
    n0 = len(core)
    for k in range(1, nr + 1):
        w = k / nr
        ring = (1 - w) * sq + w * circ
        pts.append(ring)
        loops.append(n0 + np.arange(4 * nc))
        n0 += 4 * nc
    quads, ring_of = [], []
    for i in range(nc):
        for j in range(nc):
            quads.append([cid(i, j), cid(i + 1, j), cid(i, j + 1), cid(i + 1, j + 1)])
            ring_of.append(-1)
    for k in range(nr):
        a, b = loops[k], loops[k + 1]
        for m in range(4 * nc):
            m1 = (m + 1) % (4 * nc)
            quads.append([a[m], a[m1], b[m], b[m1]])
            ring_of.append(k)
    return np.concatenate(pts), np.array(quads), np.array(ring_of)


def sizes_for(target_tets: int):
    """(nc, nr_fluid, nx) giving roughly ``target_tets`` tetrahedra with near-isotropic cells."""
    L, Rtot = X_MAX - X_MIN, R_LUMEN
    best = None
    for nc in range(2, 200, 2):
        nr = max(1, (3 * nc) // 4)
        h = 0.8 * Rtot / nc
        nx = max(2, int(round(L / h)))
        tets = 12 * nx * (nc * nc + 4 * nc * (nr + 1))
        if best is None or abs(tets - target_tets) < abs(best[0] - target_tets):
            best = (tets, nc, nr, nx)
    return best[1:]


def generate(target_tets: int = 50000, seed: int = 0, jitter: float = 0.1):
    """Returns dict(coords, tets, cell_markers, facets, facet_markers) of an offset-stenosis tube."""
    nc, nr, nx = sizes_for(target_tets)
    P, Q, ring_of = _cross_section(nc, nr)                    # fluid: the last loop is the unit circle = interface
    nloop = 4 * nc
    n_fluid_pts = len(P)
    intf = np.arange(n_fluid_pts - nloop, n_fluid_pts)        # interface loop
    wall = n_fluid_pts + np.arange(nloop)                     # outer-wall loop, same angles
    P = np.concatenate([P, P[intf]])
    m1 = (np.arange(nloop) + 1) % nloop
    Q = np.concatenate([Q, np.stack([intf, intf[m1], wall, wall[m1]], axis=1)])
    ring_of = np.concatenate([ring_of, np.full(nloop, nr)])   # ring nr = vessel wall (solid)
    npl = len(P)
    is_outer = np.zeros(npl, bool)
    is_outer[wall] = True
    is_intf = np.zeros(npl, bool)
    is_intf[intf] = True
    xs = np.linspace(X_MIN, X_MAX, nx + 1)
    scale, off = _stenosis(xs)
    coords = np.empty((nx + 1, npl, 3))
    for s in range(nx + 1):
        Rl = R_LUMEN * scale[s]
        yz = P * Rl
        yz[wall] = P[wall] * (Rl + T_WALL)
        coords[s, :, 0] = xs[s]
        coords[s, :, 1] = yz[:, 0] + off[s]
        coords[s, :, 2] = yz[:, 1]
    # jitter strictly interior fluid nodes (not on the interface / wall, not on the end planes)
    rng = np.random.default_rng(seed)
    h = min((X_MAX - X_MIN) / nx, 0.5 * 0.8 * R_LUMEN / nc)
    movable = ~(is_outer | is_intf)
    dj = jitter * h * (rng.random((nx + 1, npl, 3)) - 0.5) * 2 * 0.5
    dj[:, ~movable, :] = 0.0
    dj[0], dj[-1] = 0.0, 0.0
    coords += dj
    V0 = (nx + 1) * npl
    coords = coords.reshape(V0, 3)
    # hexahedra: (slice s, quad q) -> 8 vertices [bottom 4 | top 4], + one centre vertex each
    nq = len(Q)
    base = (np.arange(nx)[:, None] * npl)
    hexv = np.concatenate([(base[:, :, None] + Q[None, :, :]), (base[:, :, None] + npl + Q[None, :, :])], axis=2)
    hexv = hexv.reshape(nx * nq, 8)
    centre = V0 + np.arange(nx * nq)
    coords = np.concatenate([coords, coords[hexv].mean(axis=1)])
    solid_hex = np.tile(ring_of == nr, nx)
    tets = []
    for f in HEX_FACES:
        fv = hexv[:, f]                                        # cyclic order a,b,c,d around the face
        amin = np.argmin(fv, axis=1)
        rolled = np.take_along_axis(fv, (amin[:, None] + np.arange(4)[None]) % 4, axis=1)   # lowest vertex first
        tets.append(np.stack([rolled[:, 0], rolled[:, 1], rolled[:, 2], centre], axis=1))
        tets.append(np.stack([rolled[:, 0], rolled[:, 2], rolled[:, 3], centre], axis=1))
    tets = np.sort(np.concatenate(tets), axis=1)
    cell_markers = np.tile(np.where(solid_hex, 2, 1), 12)
    # all facets, markers on the boundary / interface -----------------------------------------------------------
    faces = np.sort(tets[:, [[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]]].reshape(-1, 3), axis=1)
    owner = np.repeat(np.arange(len(tets)), 4)
    Vn = len(coords)
    key = (faces[:, 0] * Vn + faces[:, 1]) * Vn + faces[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.r_[True, ks[1:] != ks[:-1]]
    idx_first = np.nonzero(first)[0]
    count = np.diff(np.r_[idx_first, len(ks)])
    facets = faces[order[idx_first]]
    c0 = owner[order[idx_first]]
    c1 = np.where(count == 2, owner[order[np.minimum(idx_first + 1, len(ks) - 1)]], -1)
    fm = np.zeros(len(facets), dtype=np.int64)
    m0 = cell_markers[c0]
    m1 = np.where(c1 >= 0, cell_markers[np.maximum(c1, 0)], 0)
    xf = coords[facets][:, :, 0]
    ext = c1 < 0
    at_in = ext & np.all(np.isclose(xf, X_MIN), axis=1)
    at_out = ext & np.all(np.isclose(xf, X_MAX), axis=1)
    fm[at_in & (m0 == 1)] = 3
    fm[at_out & (m0 == 1)] = 2
    fm[(at_in | at_out) & (m0 == 2)] = 11
    fm[ext & ~at_in & ~at_out] = 33
    fm[(c1 >= 0) & (m0 != m1)] = 22
    return dict(coords=coords, tets=tets, cell_markers=cell_markers, facets=facets, facet_markers=fm,
                shape=(nc, nr, nx))


def write_mesh(path, target_tets: int = 50000, seed: int = 0):
    """Write ``<path>`` (DOLFIN HDF5 layout) plus the two probe JSON files the problem file loads."""
    m = generate(target_tets, seed)
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    C = len(m["tets"])
    attrs = lambda ct: {"celltype": ct, "partition": np.array([0], dtype=np.uint64)}
    root = Group()
    g = Group()
    g["coordinates"] = Dataset(m["coords"])
    g["topology"] = Dataset(m["tets"].astype(np.int64), attrs("tetrahedron"))
    g["cell_indices"] = Dataset(np.arange(C, dtype=np.int64))
    b = Group()
    b["coordinates"] = Dataset(m["coords"])
    b["topology"] = Dataset(m["facets"].astype(np.int64), attrs("triangle"))
    b["values"] = Dataset(m["facet_markers"].astype(np.uint64))
    d = Group()
    d["coordinates"] = Dataset(m["coords"])
    d["topology"] = Dataset(m["tets"].astype(np.int64), attrs("tetrahedron"))
    d["cell_indices"] = Dataset(np.arange(C, dtype=np.int64))
    d["values"] = Dataset(m["cell_markers"].astype(np.uint64))
    root["mesh"], root["boundaries"], root["domains"] = g, b, d
    write_h5(path, root)
    # probes: 7 points on the axis (fluid), 50 seeded wall mid-thickness points near the stenosis (solid)
    xs = np.linspace(X_MIN + 0.002, X_MAX - 0.002, 7)
    _, off = _stenosis(xs)
    fluid = np.stack([xs, off, np.zeros(7)], axis=1)
    rng = np.random.default_rng(seed + 1)
    xw = rng.uniform(0.002, 0.014, 50)
    th = rng.uniform(0, 2 * np.pi, 50)
    sc, ow = _stenosis(xw)
    rw = R_LUMEN * sc + 0.5 * T_WALL
    solid = np.stack([xw, ow + rw * np.cos(th), rw * np.sin(th)], axis=1)
    (path.parent / (path.stem + "_probe_point.json")).write_text(json.dumps(fluid.tolist()))
    (path.parent / (path.stem + "_solid_probe.json")).write_text(json.dumps(solid.tolist()))
    return m
