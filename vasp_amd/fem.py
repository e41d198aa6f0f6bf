"""The small host API the re-hosted problem files are written against.

The reference's hooks manipulate live DOLFIN/UFL objects (``DirichletBC``, ``Measure``,
``FacetNormal``, ``F_solid_linear += ...``) [REF src/vasp/simulations/offset_stenosis.py:151-196].
DOLFIN exists neither in the build image nor on the GPU box, so the same hooks are expressed with
descriptors carrying the same information: which sub-space, which facet marker,
which values; which surface term is added to the solid form.  The time-step kernel behind the C-ABI
consumes the resolved dof lists / facet lists.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from .mesh import FIELD_D, FIELD_P, FIELD_V, FsiMesh, p2_basis


class SubSpace:
    """``DVP.sub(field)`` or ``DVP.sub(field).sub(comp)``."""

    def __init__(self, mesh: FsiMesh, fld: int, comp: Optional[int] = None):
        self.mesh, self.field, self.comp = mesh, fld, comp

    def sub(self, i: int) -> "SubSpace":
        if self.field == FIELD_P or self.comp is not None:
            raise ValueError("scalar sub-space has no components")
        return SubSpace(self.mesh, self.field, int(i))

    def ufl_element(self):
        return ("P", 1 if self.field == FIELD_P else 2, self.comp)


class MixedSpace:
    """``FunctionSpace(mesh, MixedElement([P2^3, P2^3, P1]))``: sub(0)=d, sub(1)=v, sub(2)=p."""

    def __init__(self, mesh: FsiMesh):
        self.mesh = mesh

    def sub(self, i: int) -> SubSpace:
        return SubSpace(self.mesh, (FIELD_D, FIELD_V, FIELD_P)[i])

    def dim(self) -> int:
        return self.mesh.num_dofs


class DirichletBC:
    """Topological Dirichlet condition: every dof of ``space`` in the closure of facets marked ``marker``.

    ``value``: a constant (scalar / 3-tuple) or, per component, an object with ``eval_nodes(x)``
    (e.g. ``WomersleyComponent``).  DOLFIN semantics: values of expressions are point
    evaluations at the dof coordinates; conditions are applied in list order, later ones win.
    """

    def __init__(self, space: SubSpace, value, boundaries: np.ndarray, marker: int):
        self.space, self.value, self.marker = space, value, int(marker)
        mesh = space.mesh
        fids = np.nonzero(np.asarray(boundaries) == self.marker)[0]
        if space.field == FIELD_P:
            nodes = np.unique(mesh.facets[fids].ravel())
        else:
            nodes = np.unique(mesh.facet_nodes[fids].ravel())
        self.nodes = nodes
        self.dofs = mesh.dofs(space.field, nodes, space.comp)

    def values(self) -> np.ndarray:
        """Current values, aligned with ``self.dofs``."""
        mesh, sp = self.space.mesh, self.space
        n = len(self.nodes)
        val = self.value
        if sp.field == FIELD_P or sp.comp is not None:
            if hasattr(val, "eval_nodes"):
                return np.asarray(val.eval_nodes(mesh.node_coords[self.nodes]), dtype=float)
            return np.full(n, float(val))
        if hasattr(val, "eval_nodes"):      # vector-valued expression (e.g. a parabolic inlet profile)
            return np.asarray(val.eval_nodes(mesh.node_coords[self.nodes]), dtype=float).reshape(n, 3).ravel()
        comps = list(val) if not np.isscalar(val) else [val] * 3
        out = np.empty((n, 3))
        for c in range(3):
            vc = comps[c]
            out[:, c] = vc.eval_nodes(mesh.node_coords[self.nodes]) if hasattr(vc, "eval_nodes") else float(vc)
        return out.ravel()


def resolve_bcs(bcs: Sequence[DirichletBC], ndof: int):
    """Unique Dirichlet dofs and, for every later call, their values with list-order precedence."""
    owner = -np.ones(ndof, dtype=np.int64)
    for i, bc in enumerate(bcs):
        owner[bc.dofs] = i
    dofs = np.nonzero(owner >= 0)[0]
    # per condition: which of its dofs it still owns (a later condition may have taken them) and where they sit in `dofs`;
    # the per-step evaluation then touches only the Dirichlet dofs, not a vector of the size of the problem
    own = []
    for i, bc in enumerate(bcs):
        mine = owner[bc.dofs] == i
        own.append((mine, np.searchsorted(dofs, bc.dofs[mine])))

    def values() -> np.ndarray:
        out = np.zeros(len(dofs))
        for bc, (mine, pos) in zip(bcs, own):
            out[pos] = bc.values()[mine]
        return out

    return dofs, values


class SurfacePressureTerm:
    """``P * inner(n('+'), psi('+')) * dS(marker)`` [REF offset_stenosis.py:184-190].

    '+' is the incident cell with the larger domain id (DOLFIN orients interior facets that way when
    the form carries cell markers — hence the note at REF offset_stenosis.py:189); ``n('+')`` is that
    cell's outward normal.  ``pressure`` is an object with attribute ``P`` (``InterfacePressure``).
    """

    def __init__(self, pressure, boundaries: np.ndarray, marker: int):
        self.pressure, self.boundaries, self.marker = pressure, boundaries, int(marker)

    def facets(self, mesh: FsiMesh):
        fids = np.nonzero(np.asarray(self.boundaries) == self.marker)[0]
        cells = mesh.facet_cells[fids]
        interior = cells[:, 1] >= 0
        fids, cells = fids[interior], cells[interior]      # dS: interior facets only
        m0 = mesh.cell_markers[cells[:, 0]]
        m1 = mesh.cell_markers[cells[:, 1]]
        plus = np.where(m1 > m0, cells[:, 1], cells[:, 0])
        return fids, plus


class RobinTerm:
    """turtleFSI ``robin_bc``: theta-weighted (k_s d + c_s v, psi) ds(ds_s_id) [REF aneurysm.py:73-76]."""

    def __init__(self, boundaries: np.ndarray, marker: int, k_s: float, c_s: float):
        self.boundaries, self.marker, self.k_s, self.c_s = boundaries, int(marker), float(k_s), float(c_s)


class FormTerms(list):
    """Stand-in for a UFL form that hooks extend with ``+=`` (only the surface terms are symbolic here)."""

    def __iadd__(self, other):
        if isinstance(other, (list, tuple)):
            self.extend(other)
        else:
            self.append(other)
        return self


class Function:
    """A view of one field (d, v or p) of a monolithic state vector, with point evaluation.

    The nodal array is resolved on first use: when the state lives on the device (``MixedFunction.mark_stale``), hooks
    that only need probes, cell statistics or the values on a patch never pull the whole vector over PCIe."""

    def __init__(self, mesh: FsiMesh, fld: int, nodal: Optional[np.ndarray] = None, backend=None, parent=None,
                 deepcopy: bool = False):
        self.mesh, self.field, self._nodal = mesh, fld, nodal
        self.backend = backend          # the time-step kernel holding the same state on the device (or None)
        self._parent, self._deepcopy = parent, deepcopy
        self._allow_extrapolation = False

    @property
    def nodal(self) -> np.ndarray:
        if self._nodal is None:
            arr = self.mesh.split(self._parent.vector())[(FIELD_D, FIELD_V, FIELD_P).index(self.field)]
            self._nodal = arr.copy() if self._deepcopy else arr
        return self._nodal

    def values_at_nodes(self, nodes: np.ndarray) -> np.ndarray:
        """Nodal values at P2 ``nodes`` (vertices for p): (n, 3) or (n,).  Fetched from the device when the host copy of
        the state is stale (``fsi_get_values``)."""
        nodes = np.asarray(nodes, dtype=np.int64)
        par = self._parent
        if self._nodal is None and par is not None and par.stale and hasattr(par.backend, "get_values"):
            vals = par.backend.get_values(par.which, self.mesh.dofs(self.field, nodes))
            return vals if self.field == FIELD_P else vals.reshape(len(nodes), 3)
        return self.nodal[nodes]

    def get_allow_extrapolation(self):
        return self._allow_extrapolation

    def set_allow_extrapolation(self, flag):
        self._allow_extrapolation = bool(flag)

    def value_shape(self):
        return () if self.field == FIELD_P else (3,)

    def degree(self):
        return 1 if self.field == FIELD_P else 2

    def __call__(self, x):
        cell, lam = self.mesh.locate(np.asarray(x, dtype=float)[None, :])
        if cell[0] < 0:
            raise RuntimeError("point is not inside the mesh (extrapolation not allowed)")
        c, l = int(cell[0]), lam[0]
        if self.field == FIELD_P:
            return float(l @ self.nodal[self.mesh.tets[c]])
        return p2_basis(l) @ self.nodal[self.mesh.tet_nodes[c]]


class MixedFunction:
    """``dvp_["n"]``: the monolithic vector with ``sub(i, deepcopy=True)`` → ``Function``.

    ``mark_stale()`` says the device state has moved on (after a time step); the host array is refreshed from the
    backend on the next ``vector()`` - i.e. only when a hook, a checkpoint or a visualization frame really needs it."""

    def __init__(self, mesh: FsiMesh, vector: np.ndarray, backend=None, which: str = "n"):
        self.mesh, self._x, self.backend, self.which = mesh, vector, backend, which
        self.stale = False

    def mark_stale(self):
        self.stale = self.backend is not None

    def vector(self) -> np.ndarray:
        if self.stale:
            self.backend.get_state(self.which, self._x)
            self.stale = False
        return self._x

    def sub(self, i: int, deepcopy: bool = False) -> Function:
        return Function(self.mesh, (FIELD_D, FIELD_V, FIELD_P)[i], None, backend=self.backend, parent=self,
                        deepcopy=deepcopy)
