"""Host logic of the solid cycle's exact coarse solve (vasp_amd/csrc/fsi_bcr.hip): the planner on real wall graphs, no device.

The level is solved by block cyclic reduction over breadth-first levels of the solid vertices; what the planner must deliver is
(i) a numbering in which the operator is block tridiagonal - every edge of the graph joins equal or neighbouring levels - with
blocks of one cross-section of the wall, and (ii) sizes that decide whether the method is used at all.  The arithmetic of the
reduction itself is restated in numpy on that numbering (``bcr_reference``) and solved against a dense factorisation; the HIP
kernels are held to the same dense solve on the GPU (tests/test_gpu_parity.py::test_exact_coarse_solve_*)."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from vasp_amd import capi


def wall_graph(tets):
    """Vertex graph of the SOLID cells of a generated offset-stenosis mesh: (nc, cptr, ccol) with a self loop per vertex."""
    from vasp_amd.meshgen import generate
    m = generate(tets, seed=0)
    solid = m["tets"][m["cell_markers"] == 2]
    verts = np.unique(solid)
    loc = -np.ones(len(m["coords"]), dtype=np.int64)
    loc[verts] = np.arange(len(verts))
    t = loc[solid]
    r = np.repeat(t, 4, axis=1).ravel()
    c = np.tile(t, (1, 4)).ravel()
    G = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(len(verts), len(verts)))
    G.sum_duplicates()
    G.sort_indices()
    return len(verts), G.indptr.astype(np.int64), G.indices.astype(np.int32), m["shape"]


def plan(nc, cptr, ccol):
    lib = capi.load_library()
    stats = np.zeros(8, dtype=np.int64)
    pos, level = np.empty(nc, dtype=np.int32), np.empty(nc, dtype=np.int32)
    rc = lib.fsi_bcr_plan_graph(nc, capi._ptr(cptr), capi._ptr(ccol), capi._ptr(stats), capi._ptr(pos), capi._ptr(level))
    assert rc == 0
    keys = ("usable", "blocks", "max_block", "levels", "bytes32", "bytes64", "setup_flops", "launches")
    return dict(zip(keys, (int(v) for v in stats))), pos, level


def bcr_reference(A, off):
    """Block cyclic reduction of a block-tridiagonal matrix (blocks off[k]:off[k+1]) exactly as fsi_bcr.hip schedules it:
    operators in FP32, vectors FP64.  Returns solve(rhs)."""
    K = len(off) - 1
    blk = lambda i, j: A[off[i]:off[i + 1], off[j]:off[j + 1]].copy()
    D = {k: blk(k, k) for k in range(K)}
    L = {k: (blk(k, k - 1) if k > 0 else None) for k in range(K)}
    U = {k: (blk(k, k + 1) if k + 1 < K else None) for k in range(K)}
    active, levels = list(range(K)), []
    while len(active) > 1:
        na = len(active)
        Dinv = {active[i]: np.linalg.inv(D[active[i]]) for i in range(1, na, 2)}
        fwd, bwd, newL, newU = [], [], {}, {}
        for i in range(1, na, 2):
            e, a, c = active[i], active[i - 1], (active[i + 1] if i + 1 < na else None)
            W, segs = [Dinv[e], -Dinv[e] @ L[e]], [("b", e), ("x", a)]
            if c is not None:
                W.append(-Dinv[e] @ U[e]); segs.append(("x", c))
            bwd.append((e, np.hstack(W).astype(np.float32), segs))
        for i in range(0, na, 2):
            j, l, r = active[i], (active[i - 1] if i > 0 else None), (active[i + 1] if i + 1 < na else None)
            W, segs = [], []
            newL[j] = newU[j] = None
            if l is not None:
                Gl = -L[j] @ Dinv[l]; W.append(Gl); segs.append(("b", l)); D[j] = D[j] + Gl @ U[l]
                if i >= 2:
                    newL[j] = Gl @ L[l]
            if r is not None:
                Gr = -U[j] @ Dinv[r]; W.append(Gr); segs.append(("b", r)); D[j] = D[j] + Gr @ L[r]
                if i + 2 < na:
                    newU[j] = Gr @ U[r]
            if W:
                fwd.append((j, np.hstack(W).astype(np.float32), segs))
        for i in range(0, na, 2):
            L[active[i]], U[active[i]] = newL[active[i]], newU[active[i]]
        levels.append((fwd, bwd))
        active = active[0::2]
    top = (active[0], np.linalg.inv(D[active[0]]).astype(np.float32))

    def solve(rhs):
        b, x = rhs.astype(np.float64).copy(), np.zeros(len(rhs))
        seg = lambda v, k: v[off[k]:off[k + 1]]
        for fwd, _ in levels:
            upd = {j: seg(b, j) + W.astype(np.float64) @ np.concatenate([seg(b, k) for _, k in segs]) for j, W, segs in fwd}
            for j, v in upd.items():
                b[off[j]:off[j + 1]] = v
        x[off[top[0]]:off[top[0] + 1]] = top[1].astype(np.float64) @ seg(b, top[0])
        for _, bwd in reversed(levels):
            for e, W, segs in bwd:
                x[off[e]:off[e + 1]] = W.astype(np.float64) @ np.concatenate([seg(b if s == "b" else x, k) for s, k in segs])
        return x
    return solve


@pytest.mark.parametrize("tets", [6000, 48000])
def test_planner_numbers_a_vessel_wall_block_tridiagonally(tets):
    nc, cptr, ccol, (ncs, nr, nx) = wall_graph(tets)
    st, pos, level = plan(nc, cptr, ccol)
    assert st["usable"] == 1 and st["blocks"] == level.max() + 1
    # every edge joins equal or neighbouring levels; pos is a permutation that ascends with the level
    rows = np.repeat(np.arange(nc), np.diff(cptr))
    assert np.abs(level[rows] - level[ccol]).max() <= 1
    assert np.array_equal(np.sort(pos), np.arange(nc))
    order = np.argsort(pos)
    assert np.all(np.diff(level[order]) >= 0)
    # blocks are cross-sections of the wall: two rings of the O-grid plus the cell centres between two stations
    ring = 4 * ncs
    assert st["max_block"] <= 3 * 3 * ring and st["blocks"] >= nx
    assert st["levels"] == int(np.ceil(np.log2(st["blocks"]))) and st["launches"] == 2 * st["levels"] + 1
    # operators: about five dense blocks per BFS level, FP32
    m = 3 * np.bincount(level)
    assert 2.0 * 4 * (m.astype(float) ** 2).sum() < st["bytes32"] < 8.0 * 4 * (m.astype(float) ** 2).sum()


def test_reduction_on_the_planners_numbering_solves_the_system():
    """A random positive definite 3x3-block operator on the 6 000-tet wall graph, renumbered by the planner: block tridiagonal
    (no entry outside the three block diagonals), and the reduction as scheduled (FP32 operators) solves it to FP32 accuracy."""
    nc, cptr, ccol, _ = wall_graph(6000)
    st, pos, level = plan(nc, cptr, ccol)
    rng = np.random.default_rng(0)
    nb = len(ccol)
    vals = rng.standard_normal((nb, 3, 3)) * 0.2
    B = sp.bsr_matrix((vals, ccol, cptr), shape=(3 * nc, 3 * nc)).tocsr()
    A = (B @ B.T + 4.0 * sp.identity(3 * nc)).toarray()            # SPD; pattern: two rings of the graph ...
    G = sp.csr_matrix((np.ones(nb), ccol, cptr), shape=(nc, nc))
    mask = sp.kron(G, np.ones((3, 3))).toarray() > 0
    A = np.where(mask, A, 0.0) + 30.0 * np.eye(3 * nc)              # ... cut back to the graph itself (kept definite by the shift)
    perm = np.empty(3 * nc, dtype=np.int64)
    for c in range(3):
        perm[3 * pos + c] = 3 * np.arange(nc) + c
    Ap = A[np.ix_(perm, perm)]
    m = 3 * np.bincount(level)
    off = np.concatenate([[0], np.cumsum(m)])
    K = len(m)
    for k in range(K):                                              # nothing outside the three block diagonals
        lo, hi = off[max(k - 1, 0)], off[min(k + 2, K)]
        assert not Ap[off[k]:off[k + 1], :lo].any() and not Ap[off[k]:off[k + 1], hi:].any()
    solve = bcr_reference(Ap, off)
    rhs = rng.standard_normal(3 * nc)
    x = solve(rhs)
    assert np.linalg.norm(Ap @ x - rhs) <= 2e-6 * np.linalg.norm(rhs)


def test_planner_declines_a_graph_that_is_not_a_thin_wall():
    """A 3-D block of vertices has breadth-first levels of thousands of unknowns: dense operators would cost more than the
    sweeps they replace, and the planner says so instead of building them."""
    n = 30
    idx = np.arange(n ** 3).reshape(n, n, n)
    pairs = []
    for ax in range(3):
        a = np.take(idx, np.arange(n - 1), axis=ax).ravel()
        b = np.take(idx, np.arange(1, n), axis=ax).ravel()
        pairs += [(a, b), (b, a)]
    r = np.concatenate([p[0] for p in pairs] + [np.arange(n ** 3)])
    c = np.concatenate([p[1] for p in pairs] + [np.arange(n ** 3)])
    G = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(n ** 3, n ** 3))
    G.sort_indices()
    st, _, _ = plan(n ** 3, G.indptr.astype(np.int64), G.indices.astype(np.int32))
    assert st["usable"] == 0 and st["max_block"] > 2000
