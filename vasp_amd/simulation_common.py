"""Host-side helpers shared by the re-hosted problem files.

Counterpart of [REF src/vasp/simulations/simulation_common.py]: probe files and probe printing
(:119-222), per-step flow statistics (:253-317), minimum Jacobian (:320-348) and the Fourier
interface pressure (:351-401).  The printed lines are what VaSP's tests and ``vasp-log-plotter`` parse
[REF src/vasp/postprocessing/log_plotter.py:71-82], so their text is reproduced verbatim.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from .mesh import FsiMesh
from .quadrature import tabulate_tet, tabulate_tri, tet_rule_deg6, tri_rule_deg6


def load_probe_points(mesh_path) -> np.ndarray:
    mesh_path = Path(mesh_path)
    with open(mesh_path.parent / (mesh_path.stem + "_probe_point.json")) as f:
        return np.array(json.load(f))


def load_solid_probe_points(mesh_path) -> np.ndarray:
    mesh_path = Path(mesh_path)
    with open(mesh_path.parent / (mesh_path.stem + "_solid_probe.json")) as f:
        return np.array(json.load(f))


def peval(f, x):
    """Point evaluation; a point outside the mesh yields +inf (the reference's MIN-allreduce sentinel)."""
    try:
        return np.asarray(f(x), dtype=float)
    except RuntimeError:
        return np.inf * np.ones(f.value_shape())


def _device_probe(f, probe_points):
    """(values (n,7) or None, located mask): probes evaluated by the time-step kernel (fsi_probe) at points located
    once on the host; points outside the mesh keep the reference's +inf sentinel."""
    backend = getattr(f, "backend", None)
    if backend is None or not hasattr(backend, "probe"):
        return None, None
    mesh = f.mesh
    pts = np.atleast_2d(np.asarray(probe_points, dtype=float))
    key = pts.tobytes()
    cache = getattr(mesh, "_probe_cache", {})
    if key not in cache:
        cache[key] = mesh.locate(pts)
        mesh._probe_cache = cache
    cells, bary = cache[key]
    ok = cells >= 0
    vals = np.full((len(pts), 7), np.inf)
    if ok.any():
        vals[ok] = backend.probe(cells[ok], bary[ok])
    return vals, ok


def print_probe_points(v, p, probe_points) -> None:
    v.set_allow_extrapolation(False)
    p.set_allow_extrapolation(False)
    vals, _ = _device_probe(v, probe_points)
    if vals is not None:
        for i, r in enumerate(vals):
            print(f"Probe Point {i}: Velocity: ({r[3]}, {r[4]}, {r[5]}) | Pressure: {r[6]}")
        return
    for i, point in enumerate(probe_points):
        x = [float(c) for c in np.asarray(point).tolist()]
        u_eval = peval(v, x)
        pp = peval(p, x)
        print(f"Probe Point {i}: Velocity: ({u_eval[0]}, {u_eval[1]}, {u_eval[2]}) | Pressure: {pp}")


def print_solid_probe_points(d, probe_points) -> None:
    d.set_allow_extrapolation(False)
    vals, _ = _device_probe(d, probe_points)
    if vals is not None:
        for i, r in enumerate(vals):
            print(f"Probe Point {i}: Displacement: {float(r[0]), float(r[1]), float(r[2])}")
        return
    for i, point in enumerate(probe_points):
        d_eval = peval(d, [float(c) for c in np.asarray(point).tolist()])
        print(f"Probe Point {i}: Displacement: {float(d_eval[0]), float(d_eval[1]), float(d_eval[2])}")


def _geometry(mesh: FsiMesh):
    cache = getattr(mesh, "_diag_cache", None)
    if cache is None:
        qp, qw = tet_rule_deg6()
        N, dN, L, dL = tabulate_tet(qp)
        x = mesh.coords[mesh.tets]
        Jm = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2)
        Jinv = np.linalg.inv(Jm)
        cache = dict(qw=qw, N=N, dN=dN, Jinv=Jinv)
        mesh._diag_cache = cache
    return cache


def dg0_velocity_magnitude(mesh: FsiMesh, v_nodal: np.ndarray) -> np.ndarray:
    """Cell-local L2 projection of sqrt(v.v) onto DG0 = quadrature mean over each cell (degree-6 rule)."""
    g = _geometry(mesh)
    vq = np.einsum("qa,cai->cqi", g["N"], v_nodal[mesh.tet_nodes])
    return np.einsum("q,cq->c", g["qw"], np.sqrt(np.einsum("cqi,cqi->cq", vq, vq))) * 6.0


def dg0_jacobian(mesh: FsiMesh, d_nodal: np.ndarray) -> np.ndarray:
    """Cell-local L2 projection of det(I + grad d) onto DG0."""
    g = _geometry(mesh)
    G = np.einsum("qak,ckj->cqaj", g["dN"], g["Jinv"])
    gd = np.einsum("cai,cqaj->cqij", d_nodal[mesh.tet_nodes], G)
    return np.einsum("q,cq->c", g["qw"], np.linalg.det(np.eye(3) + gd)) * 6.0


def inlet_flux(mesh: FsiMesh, v, dsi) -> float:
    """assemble(inner(v, n) * ds(inlet)); ``dsi`` = (facet ids, areas, outward normals); ``v``: a velocity ``Function``
    (only its values on the patch are fetched) or an (N2, 3) nodal array."""
    fids, area, normal = dsi
    tp, tw = tri_rule_deg6()
    Nf = tabulate_tri(tp)
    w = 2.0 * tw @ Nf                                    # ∫N_a / area
    fn = mesh.facet_nodes[fids]
    if hasattr(v, "values_at_nodes"):
        nodes, inv = np.unique(fn.ravel(), return_inverse=True)
        vf = v.values_at_nodes(nodes)[inv].reshape(len(fids), 6, 3)
    else:
        vf = np.asarray(v)[fn]
    vn = np.einsum("fai,fi->fa", vf, normal)
    return float(np.sum(area[:, None] * vn * w[None, :]))


def calculate_and_print_flow_properties(dt, mesh, v, inlet_area, mu_f, rho_f, n, dsi, local_rhs=False) -> None:
    flow_rate_inlet = abs(inlet_flux(mesh, v, dsi))
    backend = getattr(v, "backend", None)
    if backend is not None and hasattr(backend, "flow_stats"):      # DG0 projection of |v| on the device (fsi_flow_stats)
        v_mean, v_min, v_max, _ = backend.flow_stats()
    else:
        V_vector = dg0_velocity_magnitude(mesh, v.nodal)
        v_mean, v_min, v_max = V_vector.mean(), V_vector.min(), V_vector.max()
    h_min = mesh.hmin()
    diam_inlet = np.sqrt(4 * inlet_area / np.pi)
    Re_mean, Re_min, Re_max = (rho_f * vv * diam_inlet / mu_f for vv in (v_mean, v_min, v_max))
    deg = v.degree()
    CFL_mean, CFL_min, CFL_max = (vv * dt / h_min * deg for vv in (v_mean, v_min, v_max))
    print("Flow Properties:")
    print(f"  Flow Rate at Inlet: {flow_rate_inlet}")
    print(f"  Velocity (mean, min, max): {v_mean}, {v_min}, {v_max}")
    print(f"  CFL (mean, min, max): {CFL_mean}, {CFL_min}, {CFL_max}")
    print(f"  Reynolds Numbers (mean, min, max): {Re_mean}, {Re_min}, {Re_max}")


def compute_minimum_jacobian(mesh, d, local_rhs=False) -> float:
    backend = getattr(d, "backend", None)
    if backend is not None and hasattr(backend, "flow_stats"):
        min_jacobian = float(backend.flow_stats()[3])
    else:
        min_jacobian = float(np.min(dg0_jacobian(mesh, d.nodal)))
    print(f"Minimum Jacobian: {min_jacobian}")
    if min_jacobian <= 0:
        print("Warning: Negative Jacobian detected.")
    return min_jacobian


class InterfacePressure:
    """Spatially constant interface pressure from Fourier coefficients with a cosine ramp."""

    def __init__(self, t, t_ramp_start, t_ramp_end, An, Bn, period, P_mean, **kwargs):
        self.t = t
        self.t_ramp_start, self.t_ramp_end = t_ramp_start, t_ramp_end
        self.An, self.Bn = np.asarray(An, dtype=float), np.asarray(Bn, dtype=float)
        self.omega = 2.0 * np.pi / period
        self.P_mean = P_mean
        self.P = 0.0

    def update(self, t):
        self.t = t
        if t < self.t_ramp_start:
            ramp_factor = 0.0
        elif t < self.t_ramp_end:
            ramp_factor = 0.5 - 0.5 * np.cos(np.pi * (t - self.t_ramp_start) / (self.t_ramp_end - self.t_ramp_start))
        else:
            ramp_factor = 1.0
        print("ramp_factor = {} m^3/s".format(ramp_factor))
        k = np.arange(len(self.An))
        Pn = abs(np.sum((self.An - 1j * self.Bn) * np.exp(1j * k * self.omega * t)))
        self.P = ramp_factor * Pn * self.P_mean
        print("Instantaneous normal stress prescribed at the FSI interface {} Pa".format(self.P))
