"""Reference-cell quadrature and P2/P1 tabulation used by host diagnostics and uploaded to the device.

The reference forces ``parameters["form_compiler"]["quadrature_degree"] = 6``
[REF src/vasp/simulations/offset_stenosis.py:18]; FFC then asks FIAT for its default degree-6 schemes:
the 24-point Keast rule on the tetrahedron and a 12-point rule on the triangle.
"""
from __future__ import annotations

import numpy as np

from .mesh import TET_EDGES, TRI_EDGES


def tet_rule_deg6():
    """(points (24,3) on the UFC reference tet, weights (24,) with sum 1/6)."""
    P, W = [], []
    for a, w in ((0.214602871259151684, 0.039922750258167949),
                 (0.040673958534611353, 0.010077211055320643),
                 (0.322337890142275646, 0.055357181543654720)):
        b = 1.0 - 3.0 * a
        P += [(b, a, a), (a, a, a), (a, a, b), (a, b, a)]
        W += [w] * 4
    a, b, c = 0.063661001875017525, 0.269672331458315867, 0.603005664791649076
    for perm in ((b, a, a), (a, b, a), (a, a, b), (c, a, a), (a, c, a), (a, a, c),
                 (a, b, c), (b, c, a), (c, a, b), (a, c, b), (b, a, c), (c, b, a)):
        P.append(perm)
        W.append(0.048214285714285714)
    return np.array(P), np.array(W) / 6.0


def tri_rule_deg6():
    """(points (12,2) on the UFC reference triangle, weights (12,) with sum 1/2)."""
    P, W = [], []
    for a, w in ((0.063089014491502, 0.050844906370207), (0.249286745170910, 0.116786275726379)):
        b = 1.0 - 2.0 * a
        P += [(a, a), (b, a), (a, b)]
        W += [w] * 3
    a, b = 0.053145049844817, 0.310352451033784
    c = 1.0 - a - b
    P += [(a, b), (b, a), (a, c), (c, a), (b, c), (c, b)]
    W += [0.082851075618374] * 6
    return np.array(P), np.array(W) / 2.0


def tabulate_tet(points):
    """P2 values N (Q,10), reference gradients dN (Q,10,3), P1 values L (Q,4), P1 gradients dL (4,3)."""
    x = np.asarray(points)
    L = np.column_stack([1.0 - x.sum(axis=1), x])
    dL = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    i, j = TET_EDGES[:, 0], TET_EDGES[:, 1]
    N = np.concatenate([L * (2.0 * L - 1.0), 4.0 * L[:, i] * L[:, j]], axis=1)
    dNv = (4.0 * L - 1.0)[:, :, None] * dL[None, :, :]
    dNe = 4.0 * (L[:, i, None] * dL[j][None] + L[:, j, None] * dL[i][None])
    return N, np.concatenate([dNv, dNe], axis=1), L, dL


def tabulate_tri(points):
    """P2 values on the reference triangle (Q,6) in the facet-local order of ``mesh.py``."""
    x = np.asarray(points)
    L = np.column_stack([1.0 - x.sum(axis=1), x])
    i, j = TRI_EDGES[:, 0], TRI_EDGES[:, 1]
    return np.concatenate([L * (2.0 * L - 1.0), 4.0 * L[:, i] * L[:, j]], axis=1)
