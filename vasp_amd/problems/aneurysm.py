"""Aneurysm FSI problem (Robin condition on the outer wall), re-hosted without DOLFIN.

Same parameters, boundary conditions and hooks as [REF src/vasp/simulations/aneurysm.py]: single fluid region, Womersley
inlet on `inlet_id`, interface pressure on `dS(fsi_id)`, `robin_bc` with k_s = 1e5, c_s = 10 on `ds(33)` (:73-76; the
driver turns these into the surface term of turtleFSI's solid_setup), time-averaged d/u/p accumulated in `post_solve`
(:196-203) and written by `finished` (:206-222).
"""
from pathlib import Path

import numpy as np

from vasp_amd.problems import *  # noqa: F401,F403
from vasp_amd.problems import _compiler_parameters
from vasp_amd.fem import DirichletBC, SurfacePressureTerm
from vasp_amd.mesh import FsiMesh
from vasp_amd.womersley import make_womersley_bcs, compute_boundary_geometry_acrn
from vasp_amd.simulation_common import load_probe_points, print_probe_points, calculate_and_print_flow_properties, \
    InterfacePressure, compute_minimum_jacobian


def set_problem_parameters(default_variables, **namespace):
    E_s_val, nu_s_val = 1E6, 0.45
    mu_s_val = E_s_val / (2 * (1 + nu_s_val))
    lambda_s_val = nu_s_val * 2. * mu_s_val / (1. - 2. * nu_s_val)

    default_variables.update(dict(
        T=0.002, dt=0.001, theta=0.501, save_step=1, save_solution_after_tstep=951, checkpoint_step=50,
        linear_solver="mumps", atol=1e-10, rtol=1e-9, recompute=20, recompute_tstep=20,
        inlet_id=2, inlet_outlet_s_id=11, fsi_id=22, outer_id=33,
        Q_mean=1.25E-06, P_mean=11200, T_Cycle=0.951,
        rho_f=1.000E3, mu_f=3.5E-3, dx_f_id=1,
        extrapolation="laplace", extrapolation_sub_type="constant",
        rho_s=1.0E3, mu_s=mu_s_val, nu_s=nu_s_val, lambda_s=lambda_s_val, dx_s_id=2,
        k_s=[1E5], c_s=[10], ds_s_id=[33], robin_bc=True,
        folder="aneurysm_results", mesh_path="mesh/file_aneurysm.h5", FC_file="FC_MCA_10", P_FC_File="FC_Pressure",
        compiler_parameters=_compiler_parameters, save_deg=2, scale_probe=True,
    ))
    return default_variables


def get_mesh_domain_and_boundaries(mesh_path, **namespace):
    mesh = FsiMesh.read(mesh_path)
    print("=== Mesh information ===\nNumber of cells: {}\nNumber of vertices: {}".format(mesh.num_cells, mesh.num_vertices))
    return mesh, mesh.cell_markers, mesh.facet_markers


def create_bcs(t, DVP, mesh, boundaries, mu_f, fsi_id, inlet_id, inlet_outlet_s_id, psi, F_solid_linear, p_deg, FC_file,
               Q_mean, P_FC_File, P_mean, T_Cycle, **namespace):
    An, Bn = np.loadtxt(Path(__file__).parent / FC_file).T
    Cn = (An - Bn * 1j) * Q_mean
    _, tmp_center, tmp_radius, tmp_normal = compute_boundary_geometry_acrn(mesh, inlet_id, boundaries)
    inlet = make_womersley_bcs(T_Cycle, None, mu_f, tmp_center, tmp_radius, tmp_normal,
                               DVP.sub(1).sub(0).ufl_element(), Cn=Cn)
    for uc in inlet:
        uc.set_t(t)

    u_inlet = [DirichletBC(DVP.sub(1).sub(i), inlet[i], boundaries, inlet_id) for i in range(3)]
    u_inlet_s = DirichletBC(DVP.sub(1), (0.0, 0.0, 0.0), boundaries, inlet_outlet_s_id)
    d_inlet = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_id)
    d_inlet_s = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_outlet_s_id)
    bcs = u_inlet + [d_inlet, u_inlet_s, d_inlet_s]

    An_P, Bn_P = np.loadtxt(Path(__file__).parent / P_FC_File).T
    interface_pressure = InterfacePressure(t=0.0, t_ramp_start=0.0, t_ramp_end=0.2, An=An_P, Bn=Bn_P, period=T_Cycle,
                                           P_mean=P_mean, degree=p_deg)
    F_solid_linear += SurfacePressureTerm(interface_pressure, boundaries, fsi_id)

    fids = np.nonzero(boundaries == inlet_id)[0]
    area, n = mesh.facet_area_normals(fids)
    dsi = (fids, area, n)
    inlet_area = float(area.sum())
    return dict(bcs=bcs, inlet=inlet, interface_pressure=interface_pressure, F_solid_linear=F_solid_linear, n=n,
                dsi=dsi, inlet_area=inlet_area)


def initiate(mesh_path, scale_probe, mesh, **namespace):
    probe_points = load_probe_points(mesh_path)
    if scale_probe:                                   # probe file in mm [REF aneurysm.py:157-158]
        probe_points = probe_points * 0.001
    return dict(probe_points=probe_points, d_mean=np.zeros((mesh.num_nodes, 3)), u_mean=np.zeros((mesh.num_nodes, 3)),
                p_mean=np.zeros(mesh.num_vertices))


def pre_solve(t, inlet, interface_pressure, **namespace):
    for uc in inlet:
        uc.set_t(t)
        uc.scale_value = -0.5 * np.cos(np.pi * t / 0.25) + 0.5 if t < 0.25 else 1.0
    interface_pressure.update(t)
    return dict(inlet=inlet, interface_pressure=interface_pressure)


def post_solve(dvp_, n, dsi, dt, mesh, inlet_area, mu_f, rho_f, probe_points, t, save_solution_after_tstep, d_mean, u_mean,
               p_mean, **namespace):
    d = dvp_["n"].sub(0, deepcopy=True)
    v = dvp_["n"].sub(1, deepcopy=True)
    p = dvp_["n"].sub(2, deepcopy=True)
    print_probe_points(v, p, probe_points)
    calculate_and_print_flow_properties(dt, mesh, v, inlet_area, mu_f, rho_f, n, dsi)
    compute_minimum_jacobian(mesh, d)
    if t >= save_solution_after_tstep * dt:
        d_mean += d.nodal
        u_mean += v.nodal
        p_mean += p.nodal
        return dict(u_mean=u_mean, d_mean=d_mean, p_mean=p_mean)
    return None


def finished(d_mean, u_mean, p_mean, visualization_folder, save_solution_after_tstep, T, dt, **namespace):
    num_steps = T / dt - save_solution_after_tstep + 1
    if num_steps > 0:
        np.savez(Path(visualization_folder) / "mean_fields.npz", d_mean=d_mean / num_steps, u_mean=u_mean / num_steps,
                 p_mean=p_mean / num_steps)
