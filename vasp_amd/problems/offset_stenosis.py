"""Offset-stenosis FSI problem, re-hosted without DOLFIN.

Same parameters, marker rules, boundary conditions, hook names and hook order as
[REF src/vasp/simulations/offset_stenosis.py]; the hooks talk to ``vasp_amd.fem`` descriptors instead of
DOLFIN objects.  ``FC_MCA_10`` / ``FC_Pressure`` are the reference's Fourier tables (data).
"""
from pathlib import Path

import numpy as np

from vasp_amd.problems import *  # noqa: F401,F403  (default_variables + default hooks)
from vasp_amd.problems import _compiler_parameters
from vasp_amd.fem import DirichletBC, SurfacePressureTerm
from vasp_amd.mesh import FsiMesh
from vasp_amd.womersley import make_womersley_bcs, compute_boundary_geometry_acrn
from vasp_amd.simulation_common import load_probe_points, print_probe_points, \
    calculate_and_print_flow_properties, load_solid_probe_points, print_solid_probe_points, InterfacePressure, \
    compute_minimum_jacobian


def set_problem_parameters(default_variables, **namespace):
    # Lame parameters from Young's modulus / Poisson ratio  [REF offset_stenosis.py:29-34]
    E_s_val, nu_s_val = 1e6, 0.45
    mu_s_val = E_s_val / (2 * (1 + nu_s_val))
    lambda_s_val = nu_s_val * 2. * mu_s_val / (1. - 2. * nu_s_val)

    default_variables.update(dict(
        T=0.951, dt=0.001, theta=0.501, save_step=1, checkpoint_step=50,
        linear_solver="mumps", atol=1e-6, rtol=1e-6, recompute=20, recompute_tstep=20,
        inlet_id=3, inlet_outlet_s_id=11, fsi_id=22, rigid_id=11, outer_id=33,
        Q_mean=2.5e-06, P_mean=11200, T_Cycle=0.951,
        rho_f=[1.0e3, 1.0e3], mu_f=[1.5e-3, 1.0e-2], dx_f_id=[1, 1001],
        extrapolation="laplace", extrapolation_sub_type="constant",
        rho_s=1.0e3, mu_s=mu_s_val, nu_s=nu_s_val, lambda_s=lambda_s_val, dx_s_id=2,
        fsi_region=[0.008, 0, 0, 0.008],
        folder="offset_stenosis_results", mesh_path="mesh/file_stenosis.h5",
        FC_file="FC_MCA_10", P_FC_File="FC_Pressure",
        compiler_parameters=_compiler_parameters, save_deg=2,
    ))
    return default_variables


def get_mesh_domain_and_boundaries(mesh_path, fsi_region, dx_f_id, fsi_id, rigid_id, outer_id, **namespace):
    mesh = FsiMesh.read(mesh_path)
    boundaries, domains = mesh.facet_markers, mesh.cell_markers
    print("=== Mesh information ===\nNumber of cells: {}\nNumber of vertices: {}".format(
        mesh.num_cells, mesh.num_vertices))

    # FSI only inside the sphere fsi_region = [x, y, z, r]; elsewhere the wall is rigid [REF :98-112]
    centre, radius = np.array(fsi_region[:3], dtype=float), fsi_region[3]
    wall = (boundaries == fsi_id) | (boundaries == outer_id)
    outside = np.sqrt(((mesh.facet_midpoints() - centre) ** 2).sum(axis=1)) > radius
    boundaries[wall & outside] = rigid_id

    # more viscous fluid near the outlet [REF :129-138]
    x_min = 0.024
    domains[(domains == dx_f_id[0]) & (mesh.cell_midpoints()[:, 0] > x_min)] = dx_f_id[1]

    return mesh, domains, boundaries


def initiate(mesh_path, **namespace):
    return dict(probe_points=load_probe_points(mesh_path), solid_probe_points=load_solid_probe_points(mesh_path))


def create_bcs(t, DVP, mesh, boundaries, mu_f, fsi_id, inlet_id, inlet_outlet_s_id, rigid_id, psi,
               F_solid_linear, p_deg, FC_file, Q_mean, P_FC_File, P_mean, T_Cycle, **namespace):
    # flow-rate Fourier coefficients -> complex, scaled by the mean flow rate [REF :156-159]
    An, Bn = np.loadtxt(Path(__file__).parent / FC_file).T
    Cn = (An - Bn * 1j) * Q_mean
    _, tmp_center, tmp_radius, tmp_normal = compute_boundary_geometry_acrn(mesh, inlet_id, boundaries)

    # NB the reference passes the dynamic viscosity mu_f[0] in VaMPy's kinematic-viscosity slot [REF :164]
    inlet = make_womersley_bcs(T_Cycle, None, mu_f[0], tmp_center, tmp_radius, tmp_normal,
                               DVP.sub(1).sub(0).ufl_element(), Cn=Cn)
    for uc in inlet:
        uc.set_t(t)

    u_inlet = [DirichletBC(DVP.sub(1).sub(i), inlet[i], boundaries, inlet_id) for i in range(3)]
    u_inlet_s = DirichletBC(DVP.sub(1), (0.0, 0.0, 0.0), boundaries, inlet_outlet_s_id)
    d_inlet = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_id)
    d_inlet_s = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_outlet_s_id)
    d_rigid = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, rigid_id)
    bcs = u_inlet + [d_inlet, u_inlet_s, d_inlet_s, d_rigid]

    # pulsatile pressure on the FSI interface, added to the solid form [REF :181-190]
    An_P, Bn_P = np.loadtxt(Path(__file__).parent / P_FC_File).T
    interface_pressure = InterfacePressure(t=0.0, t_ramp_start=0.0, t_ramp_end=0.2, An=An_P, Bn=Bn_P,
                                           period=T_Cycle, P_mean=P_mean, degree=p_deg)
    F_solid_linear += SurfacePressureTerm(interface_pressure, boundaries, fsi_id)

    # inlet patch for the flow-rate diagnostic [REF :192-194]
    fids = np.nonzero(boundaries == inlet_id)[0]
    area, n = mesh.facet_area_normals(fids)
    dsi = (fids, area, n)
    inlet_area = float(area.sum())
    return dict(bcs=bcs, inlet=inlet, interface_pressure=interface_pressure, F_solid_linear=F_solid_linear, n=n,
                dsi=dsi, inlet_area=inlet_area)


def pre_solve(t, inlet, interface_pressure, **namespace):
    for uc in inlet:
        uc.set_t(t)
        # cosine ramp over the first 250 ms [REF :204-208]
        uc.scale_value = -0.5 * np.cos(np.pi * t / 0.25) + 0.5 if t < 0.25 else 1.0
    interface_pressure.update(t)
    return dict(inlet=inlet, interface_pressure=interface_pressure)


def post_solve(probe_points, solid_probe_points, dvp_, dt, mesh, inlet_area, dsi, mu_f, rho_f, n, **namespace):
    d = dvp_["n"].sub(0, deepcopy=True)
    v = dvp_["n"].sub(1, deepcopy=True)
    p = dvp_["n"].sub(2, deepcopy=True)

    print_probe_points(v, p, probe_points)
    print_solid_probe_points(d, solid_probe_points)
    calculate_and_print_flow_properties(dt, mesh, v, inlet_area, mu_f[0], rho_f[0], n, dsi)
    compute_minimum_jacobian(mesh, d)
