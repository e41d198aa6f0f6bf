"""Womersley inlet profile and inlet-plane geometry, DOLFIN-free.

The offset-stenosis problem builds its inlet velocity with VaMPy (un-vendored, git master
[REF pyproject.toml:19]): ``compute_boundary_geometry_acrn(mesh, inlet_id, boundaries)`` and
``make_womersley_bcs(T_Cycle, None, mu_f[0], centre, radius, normal, element, Cn=Cn)``
[REF src/vasp/simulations/offset_stenosis.py:156-170, 199-208].  This module restates the published
behaviour of those two functions:

    u(x,t) = -n_i * scale_value * Re[ 2 C_0/(pi R^2) (1 - y^2)
             + sum_n C_n/(pi R^2) (J0(b_n) - J0(b_n y)) / (J0(b_n) - 2 J1(b_n)/b_n) exp(i n w t) ],
    b_n = R sqrt(n w / nu) sqrt(i^3),  y = r/R,  r measured in the inlet plane, t taken modulo the period.

The values are needed only at the inlet P2 nodes, once per time step: host work (complex Bessel
functions via scipy), handed to the device as Dirichlet values.
"""
from __future__ import annotations

import numpy as np
from scipy.special import jv

from .mesh import FsiMesh


def compute_boundary_geometry_acrn(mesh: FsiMesh, ind: int, facet_markers=None):
    """Area, centre, radius = sqrt(A/pi) and unit (outward) normal of the boundary patch ``ind``."""
    fm = mesh.facet_markers if facet_markers is None else facet_markers
    fids = np.nonzero(fm == ind)[0]
    if len(fids) == 0:
        raise ValueError(f"no facets with marker {ind}")
    area, n = mesh.facet_area_normals(fids)
    A = float(area.sum())
    if not A > 0.0:
        raise AssertionError("Expecting positive area, probably mismatch between mesh and markers!")
    cen = mesh.coords[mesh.facets[fids]].mean(axis=1)
    c = (area[:, None] * cen).sum(axis=0) / A
    ni = (area[:, None] * n).sum(axis=0)
    normal = ni / np.sqrt((ni ** 2).sum())
    r = np.sqrt(A / np.pi)
    return A, c, r, normal


class WomersleyComponent:
    """One Cartesian component of the Womersley profile (VaMPy ``WomersleyComponent``), evaluated at nodes."""

    def __init__(self, radius, center, normal, normal_component, period, nu, Q):
        self.radius = float(radius)
        self.center = np.asarray(center, dtype=float)
        self.normal = np.asarray(normal, dtype=float)
        self.normal_component = float(normal_component)
        self.period = float(period)
        self.nu = float(nu)
        self.Qn = np.asarray(Q, dtype=complex)
        self.N = len(self.Qn)
        self.t = None
        self.scale_value = 1.0
        self.omega = 2.0 * np.pi / self.period
        self.ns = np.arange(1, self.N)
        alpha = self.radius * np.sqrt(self.ns * (self.omega / self.nu))
        self.beta = alpha * np.sqrt(1j ** 3)
        self.jn0_betas = jv(0, self.beta)
        self.jn1_betas = jv(1, self.beta)
        self._cache_key = None
        self._coeffs = None

    def set_t(self, t):
        self.t = float(t) % self.period
        self._expnt = np.exp((self.omega * self.t * 1j) * self.ns)

    def _r_dependent_coeffs(self, x):
        """Complex coefficients (n_pts, N) that depend on the radial position only."""
        key = (x.shape, x.tobytes())
        if key == self._cache_key:
            return self._coeffs
        rv = x - self.center
        rvn = rv @ self.normal
        rp = rv - rvn[:, None] * self.normal
        y = np.sqrt(np.einsum("ij,ij->i", rp, rp)) / self.radius
        pir2 = np.pi * self.radius ** 2
        co = np.zeros((len(x), self.N), dtype=complex)
        co[:, 0] = (2.0 * self.Qn[0] / pir2) * (1.0 - y ** 2)
        bn = self.beta[None, :]
        j0 = self.jn0_betas[None, :]
        j1 = self.jn1_betas[None, :]
        co[:, 1:] = (self.Qn[None, 1:] / pir2) * (j0 - jv(0, bn * y[:, None])) / (j0 - (2.0 / bn) * j1)
        self._cache_key, self._coeffs = key, co
        return co

    def eval_nodes(self, x: np.ndarray) -> np.ndarray:
        co = self._r_dependent_coeffs(np.ascontiguousarray(x, dtype=float))
        # (an explicit sum: the complex matrix-vector product of this small shape goes through a threaded BLAS path that
        # costs 30 ms per call - per component and time step)
        wom = (co[:, 0] + (co[:, 1:] * self._expnt[None, :]).sum(axis=1)).real
        return -self.normal_component * self.scale_value * wom


def make_womersley_bcs(period, Q, nu, center, radius, normal, element=None, Cn=None):
    """List of three ``WomersleyComponent`` (x, y, z), as VaMPy's function of the same name with ``Cn`` given."""
    if Cn is None:
        raise NotImplementedError("only the Cn (Fourier coefficients of the flow rate) form is used by VaSP")
    return [WomersleyComponent(radius, center, normal, nc, period, nu, Cn) for nc in normal]
