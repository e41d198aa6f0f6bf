"""Element partition of one monolithic FSI problem across the GPUs of a node.

Counterpart of what DOLFIN/PETSc do when the reference runs as ``mpirun -np N turtleFSI -p offset_stenosis ...``
(mesh distributed by cells, ghosted vectors, MPI sums in the Krylov solver and in ``norm``; the reference's own MPI
reductions at this boundary are in REF src/vasp/simulations/simulation_common.py:213-220).

Layout chosen for one process per GPU over RCCL/xGMI:

* **nodes have owners**: P2 nodes are cut into ``world`` slabs along the longest axis of the mesh, sized so that owned +
  ghost nodes per rank are equal (vessel-like domains: the cut surfaces are cross-sections, every rank has at most two neighbours, and each
  neighbour pair has its own xGMI link);
* **a rank holds every cell that touches a node it owns**, grown by ``overlap`` further node layers (its owned cells
  first, then the ghost cells), so the rows of owned nodes - and of the ghost nodes inside the overlap - are assembled
  completely on the device without any exchange; only the outermost node layer is carried as identity rows;
* per Krylov iteration there are **two owner->ghost refreshes** (the residual into the overlap, the preconditioned
  direction back; pairwise exchanges of ~10^4-10^5 doubles per neighbour) and a handful of scalar all-reduces; the
  block preconditioner is rank-local (restricted additive Schwarz on the overlap) and needs nothing else from the
  other ranks;
* the host keeps the whole mesh on every rank (a few hundred MB at 1 M tets): hooks, Dirichlet data and post-processing
  see the global problem, the device sees the local one.

Everything in ``Partition`` is plain numpy (tested on CPU, 2 gloo ranks); ``DistBackend`` binds it to ``libvaspfsi.so``
through ``fsi_set_partition`` and moves the halo with ``torch.distributed`` (backend "nccl" = RCCL on the GPUs; "gloo"
with host staging only for rehearsals with several ranks on one card).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np


def node_owners(node_coords: np.ndarray, world: int) -> np.ndarray:
    """Owner rank of every P2 node: ``world`` slabs of equal node count along the longest axis (ties by node id)."""
    n = len(node_coords)
    axis = int(np.argmax(np.ptp(node_coords, axis=0)))
    order = np.argsort(node_coords[:, axis], kind="stable")
    owner = np.empty(n, dtype=np.int32)
    owner[order] = (np.arange(n, dtype=np.int64) * world // n).astype(np.int32)
    return owner


def axis_order(node_coords: np.ndarray) -> np.ndarray:
    """Nodes sorted along the longest axis of the mesh: contiguous runs are slabs (cross-sections of a tube)."""
    axis = int(np.argmax(np.ptp(node_coords, axis=0)))
    return np.argsort(node_coords[:, axis], kind="stable")


def geodesic_order(node_coords: np.ndarray, tn: np.ndarray) -> np.ndarray:
    """Nodes sorted by their graph distance from one end of the mesh (breadth-first levels over the element graph, ties
    along the longest axis): contiguous runs are *geodesic* slabs - cross-sections that follow a vessel around bends and
    into its branches, which is what a slab along a coordinate axis is on a straight tube only.  The AVF workload
    [REF src/vasp/simulations/avf.py:55-59: artery + vein joined at the anastomosis] is not a tube; an axis slab there
    cuts both vessels at once, a geodesic slab cuts where the front is.  Components that are not connected to the
    start node follow one after the other, each from its own end."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import shortest_path
    n = len(node_coords)
    tn = np.asarray(tn, dtype=np.int64)
    ii, jj = np.triu_indices(tn.shape[1], 1)
    r, c = tn[:, ii].ravel(), tn[:, jj].ravel()
    g = sp.coo_matrix((np.ones(len(r), dtype=np.int8), (r, c)), shape=(n, n)).tocsr()
    g.data[:] = 1
    axis = int(np.argmax(np.ptp(node_coords, axis=0)))
    along = np.argsort(node_coords[:, axis], kind="stable")
    level = np.full(n, -1, dtype=np.int64)
    base = 0
    for start in along:                                   # one pass per connected component, from its lowest end
        if level[start] >= 0:
            continue
        dist = shortest_path(g, method="D", directed=False, unweighted=True, indices=int(start))
        reach = np.isfinite(dist)
        level[reach] = base + dist[reach].astype(np.int64)
        base = int(level[reach].max()) + 1
        if (level >= 0).all():
            break
    return np.lexsort((node_coords[:, axis], level))


def owners_from_order(order: np.ndarray, tn: np.ndarray, world: int, overlap: int, rounds: int = 1) -> np.ndarray:
    """``world`` contiguous runs of the node sequence ``order``, with the cuts moved so that the *local* node counts
    (owned + ghost layers: what a rank's kernels run over) are equal - end runs have one ghost side, interior runs two."""
    n = len(order)
    share = np.full(world, n / world)
    owner = np.empty(n, dtype=np.int32)
    for it in range(rounds + 1):
        bounds = np.round(np.cumsum(share)).astype(np.int64)
        bounds[-1] = n
        owner[order] = np.searchsorted(bounds, np.arange(n, dtype=np.int64), side="right").astype(np.int32)
        if it == rounds or world == 1:
            break
        ghosts = np.array([local_sets(owner, tn, q, overlap)[2].sum() for q in range(world)]) - np.bincount(owner, minlength=world)
        share = np.maximum((n + ghosts.sum()) / world - ghosts, 0.25 * n / world)
        share *= n / share.sum()
    return owner


def local_node_counts(owner: np.ndarray, tn: np.ndarray, world: int, overlap: int) -> np.ndarray:
    return np.array([int(local_sets(owner, tn, q, overlap)[2].sum()) for q in range(world)])


def balanced_owners(node_coords: np.ndarray, tn: np.ndarray, world: int, overlap: int, rounds: int = 1,
                    method: Optional[str] = None) -> np.ndarray:
    """Owner rank of every P2 node.  ``method`` (default: $VASPFSI_PARTITION, else "auto"):

    * "slab": runs of the nodes sorted along the longest axis (round 1/2; right for a tube);
    * "geodesic": runs of the nodes sorted by graph distance from one end (``geodesic_order``; bends and branches);
    * "auto": both are built and the one whose largest rank-local problem (owned + ghost nodes) is smaller wins - on a
      straight tube the two agree and the slab is kept."""
    import os
    method = method or os.environ.get("VASPFSI_PARTITION", "auto")
    if world == 1:
        return np.zeros(len(node_coords), dtype=np.int32)
    if method not in ("slab", "geodesic", "auto"):
        raise ValueError(f"VASPFSI_PARTITION={method!r}: expected slab, geodesic or auto")
    slab = owners_from_order(axis_order(node_coords), tn, world, overlap, rounds) if method != "geodesic" else None
    if method == "slab":
        return slab
    geo = owners_from_order(geodesic_order(node_coords, tn), tn, world, overlap, rounds)
    if method == "geodesic":
        return geo
    return geo if local_node_counts(geo, tn, world, overlap).max() < local_node_counts(slab, tn, world, overlap).max() else slab


def _p2_node_coords(desc: dict) -> np.ndarray:
    """Coordinates of the P2 nodes (vertices, then edge midpoints) from coords + tet_nodes alone."""
    from .mesh import TET_EDGES
    coords = np.asarray(desc["coords"], dtype=np.float64)
    tn = np.asarray(desc["tet_nodes"], dtype=np.int64)
    V, N2 = len(coords), int(desc["num_nodes"])
    out = np.empty((N2, 3))
    out[:V] = coords
    for e in range(6):                                   # edge node 4+e sits between local vertices TET_EDGES[e]
        a, b = tn[:, TET_EDGES[e, 0]], tn[:, TET_EDGES[e, 1]]
        out[tn[:, 4 + e]] = 0.5 * (coords[a] + coords[b])
    return out


def node_dofs(nodes: np.ndarray, n2: int, nv: int) -> np.ndarray:
    """User-layout dofs carried by P2 ``nodes`` of a mesh with n2 nodes / nv vertices: d (node-major), v, then p of the
    vertices among them.  Sender and receiver apply this to the same node sequence in their own numbering."""
    nodes = np.asarray(nodes, dtype=np.int64)
    d = (3 * nodes[:, None] + np.arange(3)).ravel()
    return np.concatenate([d, 3 * n2 + d, 6 * n2 + nodes[nodes < nv]])


def local_sets(owner: np.ndarray, tn: np.ndarray, rank: int, overlap: int):
    """(local cells, nodes with complete rows, local nodes) of ``rank`` as boolean masks: the cells touching an owned
    node, grown by ``overlap`` further node layers; rows are complete for the nodes all of whose cells are local."""
    n2 = len(owner)
    inner = owner == rank
    cells = inner[tn].any(axis=1)
    for _ in range(int(overlap)):
        inner = np.zeros(n2, dtype=bool)
        inner[tn[cells]] = True
        cells = inner[tn].any(axis=1)
    nodes = np.zeros(n2, dtype=bool)
    nodes[tn[cells]] = True
    return cells, inner, nodes


class Partition:
    """The part of a global problem description (``monolithic.build_description``) that one rank holds."""

    def __init__(self, desc: dict, rank: int, world: int, owner: Optional[np.ndarray] = None, overlap: int = 2, sets=None):
        self.rank, self.world, self.overlap = int(rank), int(world), int(overlap)
        tn = np.asarray(desc["tet_nodes"], dtype=np.int64)
        V, N2, Cg = len(desc["coords"]), int(desc["num_nodes"]), len(tn)
        self.V, self.N2 = V, N2
        self.owner = (balanced_owners(_p2_node_coords(desc), tn, world, overlap) if owner is None
                      else np.asarray(owner, dtype=np.int32))
        if sets is None:                                                     # (build_all_parts hands the same sets to every rank's part)
            sets = {q: local_sets(self.owner, tn, q, overlap) for q in range(world)}
        local, inner, node_mask = sets[rank]
        cell_owner = self.owner[tn[:, 0]]                                    # a cell is counted by the owner of its first vertex
        owned_cells = np.nonzero(local & (cell_owner == rank))[0]
        ghost_cells = np.nonzero(local & (cell_owner != rank))[0]
        self.cells = np.concatenate([owned_cells, ghost_cells])              # global ids, owned first
        self.num_owned_cells = len(owned_cells)
        self.nodes = np.nonzero(node_mask)[0]                                # local -> global (vertices first: ids < V)
        self.nv = int(np.searchsorted(self.nodes, V))
        self.n2 = len(self.nodes)
        self.g2l = -np.ones(N2, dtype=np.int64)
        self.g2l[self.nodes] = np.arange(self.n2)
        self.cell_g2l = -np.ones(Cg, dtype=np.int64)
        self.cell_g2l[self.cells] = np.arange(len(self.cells))
        self.ndof = 6 * self.n2 + self.nv
        self.owned_local = self.owner[self.nodes] == rank                    # (n2,) bool
        self.complete_local = inner[self.nodes]                              # rows assembled completely from the local cells

        # ---- halo lists: rank q receives its non-owned local nodes from their owners, ascending by global node id
        self.recv_nodes: Dict[int, np.ndarray] = {}                         # owner p -> global nodes I receive
        self.send_nodes: Dict[int, np.ndarray] = {}                         # rank q  -> global nodes I send
        for p in range(world):
            if p == rank:
                continue
            r = np.nonzero(node_mask & (self.owner == p))[0]
            s_ = np.nonzero(sets[p][2] & (self.owner == rank))[0]
            if len(r):
                self.recv_nodes[p] = r
            if len(s_):
                self.send_nodes[p] = s_
        dofs = lambda nodes: node_dofs(self.g2l[nodes], self.n2, self.nv)
        self.recv_counts = [len(dofs(self.recv_nodes[p])) if p in self.recv_nodes else 0 for p in range(world)]
        self.send_counts = [len(dofs(self.send_nodes[p])) if p in self.send_nodes else 0 for p in range(world)]
        cat = lambda d: (np.concatenate([dofs(d[p]) for p in sorted(d)]) if d else np.zeros(0, dtype=np.int64))
        self.ghost_dofs = cat(self.recv_nodes)                               # local user-layout dofs, in recv-buffer order
        self.send_dofs = cat(self.send_nodes)                                # ... in send-buffer order
        self.identity_dofs = node_dofs(np.nonzero(~self.complete_local)[0], self.n2, self.nv)   # outermost layer

        # ---- local <-> global dofs ------------------------------------------------------------------------------
        nd = self.nodes
        d = (3 * nd[:, None] + np.arange(3)).ravel()
        self.l2g_dofs = np.concatenate([d, 3 * N2 + d, 6 * N2 + nd[:self.nv]])
        own3 = np.repeat(self.owned_local, 3)
        self.owned_dof_mask = np.concatenate([own3, own3, self.owned_local[:self.nv]])

        # ---- the local problem description ------------------------------------------------------------------------
        ld = {k: desc[k] for k in ("fluid_props", "solid_props", "solid_models", "dt", "theta", "delta", "laplace_alpha")
              if k in desc}
        ld["coords"] = np.ascontiguousarray(np.asarray(desc["coords"])[nd[:self.nv]])
        ld["tet_nodes"] = self.g2l[tn[self.cells]]
        ld["tets"] = ld["tet_nodes"][:, :4]
        ld["num_nodes"] = self.n2
        ld["cell_kind"] = np.asarray(desc["cell_kind"])[self.cells]
        ld["cell_region"] = np.asarray(desc["cell_region"])[self.cells]
        bc = np.asarray(desc.get("bc_dofs", np.zeros(0)), dtype=np.int64)
        fld = np.minimum(bc // (3 * N2), 2)
        node = np.where(fld < 2, (bc - fld * 3 * N2) // 3, bc - 6 * N2)
        comp = np.where(fld < 2, (bc - fld * 3 * N2) % 3, 0)
        ln = self.g2l[node]
        self.bc_sel = np.nonzero(ln >= 0)[0]                                 # positions in the global Dirichlet list
        ln, f, c = ln[self.bc_sel], fld[self.bc_sel], comp[self.bc_sel]
        ld["bc_dofs"] = np.where(f < 2, f * 3 * self.n2 + 3 * ln + c, 6 * self.n2 + ln)
        for key, extra in (("pressure_facets", ("pressure_facet_cell",)), ("robin_facets", ("robin_k", "robin_c"))):
            fn = desc.get(key)
            if fn is None or not len(fn):
                continue
            fn = np.asarray(fn, dtype=np.int64)
            keep = inner[fn].any(axis=1)                                     # facets that load a row assembled here
            ld[key] = self.g2l[fn[keep]]
            for e in extra:
                ld[e] = np.asarray(desc[e])[keep]
            if key == "pressure_facets":
                ld["pressure_facet_cell"] = self.cell_g2l[np.asarray(ld["pressure_facet_cell"], dtype=np.int64)]
                if (ld["pressure_facet_cell"] < 0).any() or (ld[key] < 0).any():
                    raise AssertionError("interface facet of an owned node outside the local cells")
        self.local_desc = ld

    # ---- a part without the global mesh (driver / worker runs: rank 0 builds every rank's part and sends it) -----------------
    EXPORT_KEYS = ("rank", "world", "overlap", "V", "N2", "cells", "num_owned_cells", "nodes", "nv", "n2", "ndof", "owned_local",
                   "complete_local", "recv_counts", "send_counts", "ghost_dofs", "send_dofs", "identity_dofs", "l2g_dofs",
                   "owned_dof_mask", "bc_sel", "local_desc")

    def export(self) -> dict:
        """Everything a rank needs of its part as plain arrays - no global-size array among them except what the part itself
        holds (its cells, nodes and dofs in global numbering)."""
        return {k: getattr(self, k) for k in self.EXPORT_KEYS}

    @classmethod
    def from_export(cls, data: dict) -> "Partition":
        """The receiving side of ``export``: a Partition that never saw the global mesh.  Global -> local lookups
        (``cell_g2l`` / ``g2l`` of the full constructor) go through sorted copies of the part's own lists instead of arrays of the
        global size."""
        self = cls.__new__(cls)
        for k in cls.EXPORT_KEYS:
            setattr(self, k, data[k])
        self.owner = None
        self.cell_g2l = _SparseMap(self.cells)
        self.g2l = _SparseMap(self.nodes)
        return self

    # ---- vectors ------------------------------------------------------------------------------------------------
    def restrict(self, x_global: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(x_global)[self.l2g_dofs])

    def owned_global_dofs(self) -> np.ndarray:
        return self.l2g_dofs[self.owned_dof_mask]


class _SparseMap:
    """global id -> local index (or -1) for the ids of one part, by binary search in the sorted ids: what ``g2l[ids]`` is with
    a dense array of the global size."""

    def __init__(self, ids: np.ndarray):
        ids = np.asarray(ids, dtype=np.int64)
        self.order = np.argsort(ids, kind="stable")
        self.sorted = ids[self.order]

    def __getitem__(self, q):
        q = np.asarray(q, dtype=np.int64)
        pos = np.searchsorted(self.sorted, q)
        pos = np.minimum(pos, max(len(self.sorted) - 1, 0))
        hit = (self.sorted[pos] == q) if len(self.sorted) else np.zeros(q.shape, dtype=bool)
        return np.where(hit, self.order[pos] if len(self.sorted) else 0, -1)


def default_overlap(world: int) -> int:
    """Overlap of the restricted Schwarz preconditioner, in node layers (scan in DistBackend.__init__)."""
    import os
    return int(os.environ["VASPFSI_OVERLAP"]) if "VASPFSI_OVERLAP" in os.environ else (4 if world <= 4 else 3)


def build_all_parts(desc: dict, world: int, overlap: Optional[int] = None, owner: Optional[np.ndarray] = None) -> List[dict]:
    """Rank 0 of a driver / worker run: the exported part of every rank from the one global description it holds - the
    counterpart of DOLFIN reading the mesh once and distributing it [REF src/vasp/simulations/offset_stenosis.py:20-23: ghost
    mode shared_vertex].  Ownership and the ranks' local sets are computed once and shared by all parts."""
    overlap = default_overlap(world) if overlap is None else int(overlap)
    tn = np.asarray(desc["tet_nodes"], dtype=np.int64)
    owner = balanced_owners(_p2_node_coords(desc), tn, world, overlap) if owner is None else np.asarray(owner, dtype=np.int32)
    sets = {q: local_sets(owner, tn, q, overlap) for q in range(world)}
    return [Partition(desc, q, world, owner, overlap, sets=sets).export() for q in range(world)]


class DistBackend:
    """The backend protocol of ``monolithic.run`` / ``bench.py`` on an element partition: one ``HipBackend`` per rank on
    the local cells, coupled through ``fsi_set_partition``.  Takes and returns *global* vectors at the host boundary."""

    _CB_RED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)
    _CB_HALO = C.CFUNCTYPE(C.c_int, C.c_void_p)

    def __init__(self, desc: Optional[dict], dist, device: int = 0, rank: Optional[int] = None, world: Optional[int] = None,
                 owner: Optional[np.ndarray] = None, overlap: Optional[int] = None, part: Optional[Partition] = None,
                 control=None, **kw):
        """``desc``: the global problem description (every rank builds its own part from it: the symmetric mode of rounds 1-3),
        or None with ``part`` = this rank's ready Partition (driver / worker mode: only rank 0 ever held the global mesh).
        ``control``: a ``ControlChannel`` - on rank 0 every collective method first tells the workers what to call."""
        import torch
        from .capi import HipBackend, _ptr
        self.torch, self.dist = torch, dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.on_gpu_wire = dist.get_backend() == "nccl"                       # RCCL moves device buffers directly
        self.control = control
        self._pending_bc = self._pending_P = None
        self._flow_stats = None
        self._flow_stats_valid = False           # DistBackend's own: cleared by newton_solve / set_state / shift, set by flow_stats
        self._inject_counts: Dict[str, int] = {}
        self._aborted = False
        # Overlap of the restricted Schwarz preconditioner, in node layers.  Measured on the 1.12 M-tet bench mesh (5 steps from
        # rest, Krylov iterations; one context: 114): 2 ranks 138 / 124 / 118 with 2 / 3 / 4 layers (3.8 / 5.1 / 6.4 % ghost
        # cells), 4 ranks 140 / 127 / 120 / 115 with 2 / 3 / 4 / 6 layers (11 / 14 / 18 / 27 %): four layers keep the count within
        # 5 % of one context; iterations x local cells is smallest at 4 layers up to 4 ranks and flat between 3 and 4 beyond
        # (profiles/r03_overlap_scan.txt).
        if overlap is None:
            overlap = default_overlap(self.world)
        self.part = part if part is not None else Partition(desc, self.rank, self.world, owner, overlap)
        self.ndof_global = 6 * self.part.N2 + self.part.V
        self.hb = HipBackend(self.part.local_desc, device=device, num_owned_cells=self.part.num_owned_cells, **kw)
        self.lib, self.ctx, self.ndof = self.hb.lib, self.hb.ctx, self.ndof_global
        dev = torch.device("cuda", device)
        self.dev = dev
        p = self.part
        self.sendbuf = torch.zeros(max(1, len(p.send_dofs)), dtype=torch.float64, device=dev)
        self.recvbuf = torch.zeros(max(1, len(p.ghost_dofs)), dtype=torch.float64, device=dev)
        self._red = torch.zeros(1024, dtype=torch.float64, device=dev)
        self._error: Optional[BaseException] = None
        self._g2l_owned: Optional[np.ndarray] = None
        self._cb_red = self._CB_RED(self._allreduce)
        self._cb_halo = self._CB_HALO(self._halo)

        class FsiComm(C.Structure):
            _fields_ = [("user", C.c_void_p), ("allreduce_sum", self._CB_RED), ("halo_exchange", self._CB_HALO)]
        self._comm = FsiComm(None, self._cb_red, self._cb_halo)
        gd = np.ascontiguousarray(p.ghost_dofs, dtype=np.int64)
        sd = np.ascontiguousarray(p.send_dofs, dtype=np.int64)
        idn = np.ascontiguousarray(p.identity_dofs, dtype=np.int64)
        self.hb._check(self.lib.fsi_set_partition(self.ctx, p.num_owned_cells, len(gd), _ptr(gd), len(idn), _ptr(idn),
                                                  len(sd), _ptr(sd),
                                                  C.c_void_p(self.sendbuf.data_ptr()), C.c_void_p(self.recvbuf.data_ptr()),
                                                  C.byref(self._comm)))
        self.history = self.hb.history
        # The library issues the collectives itself on its solver stream (fsi_set_rccl): no callback, no host staging, no
        # device synchronise per exchange.  Default on the nccl backend with more than one rank (one rank per GPU: RCCL
        # refuses two ranks on one device); VASPFSI_RCCL=0 keeps the torch.distributed callbacks above, =1 asks for the
        # library's transport whatever the world size and raises if it cannot be had.  Every step of the set-up is agreed
        # across the ranks (a rank that cannot load RCCL or fails in ncclCommInitRank takes all ranks back to the callbacks).
        self.library_rccl = False
        want = os.environ.get("VASPFSI_RCCL", "auto")
        if want == "1" and not self.on_gpu_wire:
            raise RuntimeError("VASPFSI_RCCL=1 needs the nccl backend (one rank per GPU)")
        if want == "1" or (want not in ("0", "off") and self.on_gpu_wire and self.world > 1):
            self.library_rccl, why = self._setup_library_rccl()
            if not self.library_rccl:
                if want == "1":
                    raise RuntimeError(f"VASPFSI_RCCL=1: {why}")
                if self.rank == 0:
                    print(f"vasp_amd.partition: library-side RCCL not available ({why}); collectives go through torch.distributed",
                          flush=True)

    def _setup_library_rccl(self):
        """fsi_rccl_unique_id on rank 0 -> broadcast -> fsi_set_rccl on every rank, each step agreed over the process group.
        Returns (ok, reason).  On any rank's failure every rank calls fsi_set_rccl(ctx, NULL, ...) and keeps the callbacks."""
        from .capi import _ptr
        torch, dist, p, dev = self.torch, self.dist, self.part, self.dev

        def all_ok(flag: bool) -> bool:
            t = torch.tensor([0.0 if flag else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            return float(t.item()) == 0.0

        ident = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = True
        if self.rank == 0:
            buf = (C.c_char * 128)()
            ok = self.lib.fsi_rccl_unique_id(buf) == 0
            if ok:
                ident.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
        if not all_ok(ok):
            return False, "librccl could not be loaded by the library (dlopen)"
        dist.broadcast(ident, src=0)
        raw = bytes(ident.cpu().numpy().tobytes())
        sc = np.ascontiguousarray(p.send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(p.recv_counts, dtype=np.int64)
        rcode = self.lib.fsi_set_rccl(self.ctx, C.c_char_p(raw), self.rank, self.world, _ptr(sc), _ptr(rc))
        msg = self.lib.fsi_last_error(self.ctx).decode() if rcode != 0 else ""
        if not all_ok(rcode == 0):
            self.lib.fsi_set_rccl(self.ctx, None, self.rank, self.world, _ptr(sc), _ptr(rc))
            return False, msg or "fsi_set_rccl failed on another rank"
        return True, ""

    # ---- transport (called back from inside fsi_solve / fsi_newton_solve) ------------------------------------------
    def _abort_transport(self, e: BaseException) -> None:
        """A collective failed on THIS rank inside a solve: the library returns FSI_ERR_DEVICE here at once, but the other ranks
        sit in the matching all_to_all / all_reduce.  Tear the job's transport down so that their pending collective fails too and
        they leave the solve with FSI_ERR_DEVICE (their callbacks' except branch) instead of waiting for a rank that has left -
        the callback path's counterpart of ncclCommAbort in csrc/fsi_rccl.hip."""
        if getattr(self, "_aborted", False):
            return
        self._aborted = True
        import sys
        print(f"vasp_amd.partition: rank {self.rank}: a collective failed inside a solve ({type(e).__name__}: {e}); "
              f"aborting the process group so that no rank waits for this one", file=sys.stderr, flush=True)
        try:
            if self.on_gpu_wire:
                from torch.distributed.distributed_c10d import _abort_process_group
                _abort_process_group()
            else:
                self.dist.destroy_process_group()          # gloo: closes the pairs' sockets; the peers' pending operation raises
        except Exception as e2:                            # noqa: BLE001 - nothing more can be done from inside a callback
            print(f"vasp_amd.partition: rank {self.rank}: aborting the process group failed too ({e2})", file=sys.stderr, flush=True)

    def _inject(self, what: str) -> None:
        """Test hook: VASPFSI_TEST_FAIL_<WHAT>_AT=k makes the k-th <what> callback of this process raise before it communicates."""
        at = os.environ.get(f"VASPFSI_TEST_FAIL_{what}_AT")
        if at is None:
            return
        n = self._inject_counts[what] = self._inject_counts.get(what, 0) + 1
        if n == int(at):
            raise RuntimeError(f"injected {what.lower()} failure (callback {n})")

    def _allreduce(self, _user, vals, n):
        try:
            self._inject("ALLREDUCE")
            torch, a = self.torch, np.ctypeslib.as_array(vals, shape=(n,))
            if self.on_gpu_wire:
                t = self._red[:n] if n <= len(self._red) else torch.empty(n, dtype=torch.float64, device=self.dev)
                t.copy_(torch.from_numpy(a))
                self.dist.all_reduce(t)
                a[:] = t.cpu().numpy()
            else:
                self.dist.all_reduce(torch.from_numpy(a))
            return 0
        except BaseException as e:          # never unwind through the C frames
            self._error = self._error or e
            self._abort_transport(e)
            return 1

    def _halo(self, _user):
        try:
            self._inject("HALO")
            torch, p = self.torch, self.part
            ns, nr = len(p.send_dofs), len(p.ghost_dofs)
            if self.on_gpu_wire:
                self.dist.all_to_all_single(self.recvbuf[:nr], self.sendbuf[:ns], p.recv_counts, p.send_counts)
            else:                            # rehearsal transport: stage through the host
                r = torch.empty(nr, dtype=torch.float64)
                self.dist.all_to_all_single(r, self.sendbuf[:ns].cpu(), p.recv_counts, p.send_counts)
                self.recvbuf[:nr].copy_(r)
            torch.cuda.synchronize(self.dev)
            return 0
        except BaseException as e:
            self._error = self._error or e
            self._abort_transport(e)
            return 1

    def _check(self, rc):
        if rc != 0 and self._error is not None:
            e, self._error = self._error, None
            raise e
        self.hb._check(rc)

    # ---- driver / worker runs: rank 0 tells the other ranks which collective method comes next -------------------------------
    def _tell(self, op: str, *args, **kw):
        """Rank 0 of a driver / worker run: announce the call (with the boundary data set since the last one) so that the
        workers' ``serve`` loops make the same call; a no-op in the symmetric mode and on the workers themselves."""
        if self.control is None or self.rank != 0:
            return
        pending = {}
        if self._pending_bc is not None:
            pending["bc"] = self._pending_bc
        if self._pending_P is not None:
            pending["P"] = self._pending_P
        self._pending_bc = self._pending_P = None
        self.control.tell((op, args, kw, pending))

    # ---- backend protocol -------------------------------------------------------------------------------------------
    def set_dirichlet_values(self, values):
        values = np.asarray(values, dtype=np.float64)
        if self.control is not None and self.rank == 0:
            self._pending_bc = values                # travels with the next collective call
        self.hb.set_dirichlet_values(values[self.part.bc_sel])

    def set_interface_pressure(self, P):
        if self.control is not None and self.rank == 0:
            self._pending_P = float(P)
        self.hb.set_interface_pressure(P)

    def newton_solve(self, **kw):
        self._flow_stats_valid = False
        self._tell("newton_solve", **{k: v for k, v in kw.items() if k != "log"})       # (the log callable stays on rank 0)
        try:
            return self.hb.newton_solve(**kw)
        except Exception:
            if self._error is not None:
                e, self._error = self._error, None
                raise e
            raise

    def shift(self):
        self._flow_stats_valid = False
        self._tell("shift")
        self.hb.shift()

    def assemble_residual(self):
        self._tell("assemble_residual")
        return self.hb.assemble_residual()

    def assemble_jacobian(self):
        self._tell("assemble_jacobian")
        self.hb.assemble_jacobian()

    def solve(self, *a, **kw):
        self._tell("solve", *a, **kw)
        return self.hb.solve(*a, **kw)

    def set_chebyshev(self, **kw):
        self._tell("set_chebyshev", **kw)
        self.hb.set_chebyshev(**kw)

    def solver_events(self):
        return self.hb.solver_events()         # the library agrees on these events across ranks: rank 0's counts are the job's

    def timers(self, reset=False):
        if reset:
            self._tell("timers", reset=True)
        return self.hb.timers(reset)

    # ---- post_solve diagnostics on the device (simulation_common picks them up by name, as for HipBackend) ---------------
    # Without them every rank would fall back to the numpy versions on the gathered global state: seconds per step.
    def _reduce(self, values, op: str):
        """All-reduce a few host scalars (op: "sum" | "min" | "max") over the job; returns a numpy array."""
        torch = self.torch
        t = torch.tensor(np.asarray(values, dtype=np.float64), dtype=torch.float64,
                         device=self.dev if self.on_gpu_wire else "cpu")
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX}[op])
        return t.cpu().numpy()

    def barrier(self):
        """Job-wide barrier + device synchronise (the brackets of bench.py's timed region); every rank remembers when it left
        its last two barriers, so that a worker knows the length of a timed region it did not drive."""
        import time
        self._tell("barrier")
        self.dist.barrier()
        if self.torch.cuda.is_available():
            self.torch.cuda.synchronize(self.dev)
        self._barrier_times = (getattr(self, "_barrier_times", (0.0, 0.0))[1], time.perf_counter())

    def aggregate(self, elapsed_s: Optional[float] = None, units: float = 0.0):
        """(max over ranks of the elapsed time, sum over ranks of the units): bench.py's measurement protocol.  A worker
        contributes the time between its last two barriers and no units."""
        from .dist import aggregate
        self._tell("aggregate")
        if elapsed_s is None:
            t = getattr(self, "_barrier_times", (0.0, 0.0))
            elapsed_s = t[1] - t[0]
        return aggregate(self.dist, elapsed_s, units, device=self.dev if self.on_gpu_wire else "cpu")

    def total_local_cells(self) -> int:
        """Cells held by all ranks together, ghost layers included."""
        self._tell("total_local_cells")
        return int(round(float(self._reduce([float(len(self.part.cells))], "sum")[0])))

    def agree_flags(self, flags):
        """Job-wide OR of rank-local booleans (the time loop's stop / pause controls): every rank gets the same answer."""
        from .dist import agree_flags
        self._tell("agree_flags", [False] * len(flags))          # workers have no flags of their own: they follow rank 0
        return agree_flags(self.dist, flags, device=self.dev if self.on_gpu_wire else "cpu")

    def flow_stats(self):
        """HipBackend.flow_stats over the whole job: every rank contributes the cells it owns (fsi_flow_stats counts those)."""
        if self._flow_stats_valid:             # second request of a step.  The flag is DistBackend's own (cleared by newton_solve /
            return self._flow_stats            # set_state / shift, set after the reductions): a rank without owned cells never
                                               # calls hb.flow_stats(), so hb's cache says nothing about it
        self._tell("flow_stats")
        n = self.part.num_owned_cells
        mean, mn, mx, mj = self.hb.flow_stats() if n > 0 else (0.0, np.inf, -np.inf, np.inf)
        tot = self._reduce([mean * n, float(n)], "sum")
        lo = self._reduce([mn, mj], "min")
        hi = self._reduce([mx], "max")
        self._flow_stats = (float(tot[0] / max(tot[1], 1.0)), float(lo[0]), float(hi[0]), float(lo[1]))
        self._flow_stats_valid = True
        return self._flow_stats

    def probe(self, cells, bary):
        """HipBackend.probe for global cell ids: the owner of a cell evaluates its points, the rest add zeros."""
        cells = np.asarray(cells, dtype=np.int64)
        bary = np.asarray(bary, dtype=np.float64).reshape(-1, 4)
        self._tell("probe", cells, bary)
        loc = self.part.cell_g2l[cells]
        mine = (loc >= 0) & (loc < self.part.num_owned_cells)
        out = np.zeros((len(cells), 7))
        if mine.any():
            out[mine] = self.hb.probe(loc[mine], bary[mine])
        return self._reduce(out.ravel(), "sum").reshape(-1, 7)

    def get_values(self, which, dofs):
        """state[dofs] for global user-layout dofs, from their owners."""
        dofs = np.asarray(dofs, dtype=np.int64)
        self._tell("get_values", which, dofs)
        if self._g2l_owned is None:                    # global dof -> local index of the dofs this rank owns (-1 elsewhere)
            own = np.nonzero(self.part.owned_dof_mask)[0]
            self._g2l_owned = (_SparseMap(self.part.l2g_dofs[own]), own)
        m, own = self._g2l_owned
        hit = m[dofs]
        loc = np.where(hit >= 0, own[np.maximum(hit, 0)], -1)
        mine = loc >= 0
        out = np.zeros(len(dofs))
        if mine.any():
            out[mine] = self.hb.get_values(which, loc[mine])
        return self._reduce(out, "sum")

    def set_state(self, which, x_global):
        self._flow_stats_valid = False
        self._tell("set_state", which, x_global)
        self.hb.set_state(which, self.part.restrict(x_global))

    def get_state(self, which, out=None):
        """Global vector assembled from the owners' entries (an all-gather over the host; not on the hot path)."""
        torch, p = self.torch, self.part
        self._tell("get_state", which)
        loc = self.hb.get_state(which)
        mine = torch.from_numpy(np.ascontiguousarray(loc[p.owned_dof_mask]))
        idx = torch.from_numpy(np.ascontiguousarray(p.owned_global_dofs()))
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        n_here = torch.tensor([len(idx)], dtype=torch.int64)
        if self.on_gpu_wire:
            sizes = [s.to(self.dev) for s in sizes]
            n_here = n_here.to(self.dev)
        self.dist.all_gather(sizes, n_here)
        sizes = [int(s.item()) for s in sizes]
        out = np.zeros(self.ndof_global) if out is None else out
        for r in range(self.world):                        # one broadcast per owner: simple, and off the hot path
            v = mine.clone() if r == self.rank else torch.empty(sizes[r], dtype=torch.float64)
            i = idx.clone() if r == self.rank else torch.empty(sizes[r], dtype=torch.int64)
            if self.on_gpu_wire:
                v, i = v.to(self.dev), i.to(self.dev)
            self.dist.broadcast(v, src=r)
            self.dist.broadcast(i, src=r)
            out[i.cpu().numpy()] = v.cpu().numpy()
        return out

    def close(self):
        self._tell("close")
        self.hb.close()


# ---------------------------------------------------------------------------------------------------------------------
# driver / worker runs (round 4; VERDICT r3 missing 5): only rank 0 reads the mesh and runs the problem file's hooks
# ---------------------------------------------------------------------------------------------------------------------
class ControlChannel:
    """Small host-side messages from rank 0 to the other ranks: which collective backend method to call next, with its
    arguments.  Goes over a gloo group of its own (object broadcasts over the nccl group would stage every message through the
    GPU); the data path - halo exchanges, all-reduces - stays on the job's main group."""

    def __init__(self, dist):
        self.dist = dist
        self.group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else None      # collective: every rank calls it

    def tell(self, msg):
        self.dist.broadcast_object_list([msg], src=0, group=self.group)

    def listen(self):
        box = [None]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def scatter(self, items):
        """items: a list with one object per rank on rank 0, None elsewhere; returns this rank's object."""
        out = [None]
        self.dist.scatter_object_list(out, items, src=0, group=self.group)
        return out[0]


def start_driver(desc: dict, dist, device: int = 0, overlap: Optional[int] = None, **kw) -> "DistBackend":
    """Rank 0 of a driver / worker run: cut the global problem into every rank's part, send the parts out, keep part 0.
    The returned backend tells the workers about every collective call it makes; ``close()`` ends their ``serve`` loops."""
    control = ControlChannel(dist)
    parts = build_all_parts(desc, dist.get_world_size(), overlap)
    mine = control.scatter([dict(part=p, kw=kw) for p in parts])
    return DistBackend(None, dist, device=device, part=Partition.from_export(mine["part"]), control=control, **kw)


def run_worker(dist, device: int = 0) -> None:
    """Ranks 1 .. N-1 of a driver / worker run: receive this rank's part - the only mesh data the process ever holds -, bind it
    to the device, then make the calls rank 0 announces until it closes the backend.  No problem file, no hooks, no output
    here: those are rank 0's, as the reference guards its prints with ``MPI.rank == 0``."""
    control = ControlChannel(dist)
    mine = control.scatter(None)
    db = DistBackend(None, dist, device=device, part=Partition.from_export(mine["part"]), **mine["kw"])
    serve(db, control)


# Backend calls whose FsiError the library agrees on across the ranks (a linear or Newton failure is all-reduced before any rank
# returns, fsi_set_partition / test_a_rank_local_failure_is_raised_on_every_rank_instead_of_hanging): rank 0 raises the same error
# from the same call and decides what happens next, so a worker may keep listening after one of these - and after nothing else.
AGREED_FAILURES = ("newton_solve", "solve", "assemble_jacobian")


def serve(db: "DistBackend", control: ControlChannel) -> None:
    """The worker's loop.  An FsiError from one of AGREED_FAILURES is rank 0's to handle (it sees the same one).  ANY other
    exception is rank-local - a Python error in probe / get_values / set_state, an out-of-memory while a message is unpickled, an
    FsiError from a call that is not collective inside the library - and rank 0 is then blocked in the matching all-reduce,
    all-gather or RCCL call for ever: the worker logs it, closes its backend and leaves the process with a non-zero status, so
    that the launcher (torch.distributed.run) tears the whole job down instead of letting it hang."""
    from .capi import FsiError
    while True:
        op = "listen"
        try:
            op, args, kw, pending = control.listen()
            if "bc" in pending:
                db.set_dirichlet_values(pending["bc"])
            if "P" in pending:
                db.set_interface_pressure(pending["P"])
            if op == "close":
                db.close()
                return
            if op == "flow_stats":
                db._flow_stats_valid = False   # rank 0 announced the call, so it is making the reductions: take part
            getattr(db, op)(*args, **kw)
        except FsiError as e:
            if op in AGREED_FAILURES:          # every rank got it; rank 0 decides (refresh and retry, stop, re-raise)
                continue
            _worker_abort(db, op, e)
        except Exception as e:                 # noqa: BLE001 - anything rank-local ends this process
            _worker_abort(db, op, e)


def _worker_abort(db: "DistBackend", op: str, e: BaseException) -> None:
    import sys
    import traceback
    print(f"vasp_amd.partition: worker rank {db.rank} failed in '{op}': {type(e).__name__}: {e}; leaving the job "
          f"(the launcher ends the other ranks)", file=sys.stderr, flush=True)
    traceback.print_exc()
    try:
        db.hb.close()
    except Exception:                          # noqa: BLE001 - the context may be what failed
        pass
    sys.stderr.flush()
    os._exit(WORKER_EXIT_CODE)                 # a plain exit of THIS process (no re-exec); skips the process group's teardown,
                                               # which would wait for the ranks that are blocked in a collective


WORKER_EXIT_CODE = 17
