"""Arteriovenous-fistula FSI problem, re-hosted without DOLFIN.

Same parameters, marker rules, boundary conditions and hooks as [REF src/vasp/simulations/avf.py]: two solid regions
(artery 2 / vein 1002) with MooneyRivlin constants (:71-80), list-valued fsi / rigid / outer ids (:55-59), sphere-based
relabelling to rigid walls (:97-138), two parabolic inlets driven by patient-specific velocity tables interpolated to
T/dt samples (:141-185, 237-253), tabulated wall pressure with a cosine ramp (:188-215), Robin condition on both outer
walls (:82-84).
"""
import numpy as np

from vasp_amd.problems import *  # noqa: F401,F403
from vasp_amd.problems import _compiler_parameters
from vasp_amd.fem import DirichletBC, SurfacePressureTerm
from vasp_amd.mesh import FsiMesh
from vasp_amd.simulation_common import load_probe_points, print_probe_points, calculate_and_print_flow_properties, \
    compute_minimum_jacobian


def set_problem_parameters(default_variables, **namespace):
    E_s_val_artery, E_s_val_vein, nu_s_val = 1E6, 1E6, 0.45
    mu_s_val_artery = E_s_val_artery / (2 * (1 + nu_s_val))
    mu_s_val_vein = E_s_val_vein / (2 * (1 + nu_s_val))
    lambda_s_val_artery = nu_s_val * 2. * mu_s_val_artery / (1. - 2. * nu_s_val)
    lambda_s_val_vein = nu_s_val * 2. * mu_s_val_vein / (1. - 2. * nu_s_val)

    default_variables.update(dict(
        T=3, dt=0.0001, theta=0.501, save_step=1, checkpoint_step=500,
        linear_solver="mumps", atol=1e-7, rtol=1e-7, recompute=30, recompute_tstep=10,
        inlet_id1=3, inlet_id2=2, outlet_id1=4, rigid_id=[11, 1011], fsi_id=[22, 1022], outlet_s_id=44,
        outer_id=[33, 1033], ds_s_id=[33, 1033],
        vel_t_ramp=0.2, p_t_ramp_start=0.05, p_t_ramp_end=0.2,
        rho_f=1.025E3, mu_f=3.5E-3, dx_f_id=1,
        extrapolation="laplace", extrapolation_sub_type="constant",
        rho_s=[1.0E3, 1.0E3], mu_s=[mu_s_val_artery, mu_s_val_vein], nu_s=nu_s_val,
        lambda_s=[lambda_s_val_artery, lambda_s_val_vein], material_model="MooneyRivlin", dx_s_id=[2, 1002],
        solid_properties=[{"dx_s_id": 2, "material_model": "MooneyRivlin", "rho_s": 1.0E3, "mu_s": mu_s_val_artery,
                           "lambda_s": lambda_s_val_artery, "C01": 0.03e6, "C10": 0.0, "C11": 2.2e6},
                          {"dx_s_id": 1002, "material_model": "MooneyRivlin", "rho_s": 1.0E3, "mu_s": mu_s_val_vein,
                           "lambda_s": lambda_s_val_vein, "C01": 0.003e6, "C10": 0.0, "C11": 0.538e6}],
        robin_bc=True, k_s=1E5, c_s=1E1,
        fsi_region=[0.33642, 0.0873934, 0.0369964, 0.002],
        mesh_path="mesh/avf.h5", patient_data_path="avf.csv", folder="avf_results",
        compiler_parameters=_compiler_parameters, save_deg=2, scale_probe=True,
    ))
    return default_variables


def get_mesh_domain_and_boundaries(mesh_path, fsi_region, fsi_id, rigid_id, outer_id, **namespace):
    mesh = FsiMesh.read(mesh_path)
    boundaries, domains = mesh.facet_markers, mesh.cell_markers
    centre, radius = np.array(fsi_region[:3], dtype=float), fsi_region[3]
    outside = np.sqrt(((mesh.facet_midpoints() - centre) ** 2).sum(axis=1)) > radius
    orig = boundaries.copy()
    for k in (0, 1):              # fsi and outer surfaces outside the sphere become the rigid wall of their region
        boundaries[outside & ((orig == fsi_id[k]) | (orig == outer_id[k]))] = rigid_id[k]
    return mesh, domains, boundaries


class VelInPara:
    """Parabolic profile -n v(t) (1 - r^2/R^2) with v(t) from the interpolated patient table and a cosine ramp."""

    def __init__(self, t, dt, vel_t_ramp, n, dsi, mesh, interp_velocity, **kwargs):
        self.t, self.dt, self.t_ramp, self.interp_velocity, self.n = t, dt, vel_t_ramp, interp_velocity, np.asarray(n)
        self.number = int(self.t / self.dt)
        fids, area, _ = dsi
        self.A = float(area.sum())
        self.c = (area[:, None] * mesh.coords[mesh.facets[fids]].mean(axis=1)).sum(axis=0) / self.A
        self.r = np.sqrt(self.A / np.pi)

    def update(self, t):
        self.t = t
        if self.number + 1 < len(self.interp_velocity):
            self.number = int(self.t / self.dt)

    def eval_nodes(self, x):
        fact_r = 1.0 - ((x - self.c) ** 2).sum(axis=1) / self.r ** 2
        fact = self.interp_velocity[self.number]
        if (self.t < self.t_ramp) and (self.t_ramp > 0.0):
            fact = fact * (-0.5 * np.cos((np.pi / self.t_ramp) * self.t) + 0.5)
        return -self.n[None, :] * fact * fact_r[:, None]


class InnerP:
    def __init__(self, t, dt, p_t_ramp_start, p_t_ramp_end, interp_P, **kwargs):
        self.t, self.dt, self.interp_P = t, dt, interp_P
        self.number = int(self.t / self.dt)
        self.p_t_ramp_start, self.p_t_ramp_end = p_t_ramp_start, p_t_ramp_end

    def update(self, t):
        self.t = t
        if self.number + 1 < len(self.interp_P):
            self.number = int(self.t / self.dt)

    @property
    def P(self):
        if self.t < self.p_t_ramp_start:
            return 0.0
        if self.t < self.p_t_ramp_end:
            return self.interp_P[self.number] * (-0.5 * np.cos((np.pi / (self.p_t_ramp_end - self.p_t_ramp_start))
                                                                * (self.t - self.p_t_ramp_start)) + 0.5)
        return self.interp_P[self.number]


def _inlet(mesh, boundaries, marker):
    fids = np.nonzero(boundaries == marker)[0]
    area, n = mesh.facet_area_normals(fids)
    ni = (area[:, None] * n).sum(axis=0)
    return (fids, area, n), ni / np.sqrt((ni ** 2).sum())


def create_bcs(DVP, mesh, boundaries, T, dt, fsi_id, inlet_id1, inlet_id2, rigid_id, psi, F_solid_linear, vel_t_ramp,
               p_t_ramp_start, p_t_ramp_end, p_deg, v_deg, patient_data_path, **namespace):
    print("Create bcs")
    dsi1, normal1 = _inlet(mesh, boundaries, inlet_id1)
    dsi2, normal2 = _inlet(mesh, boundaries, inlet_id2)

    # patient data: columns PA velocity, DA velocity, venous pressure; first row is a header [REF avf.py:237-253]
    patient_data = np.loadtxt(patient_data_path, skiprows=1, delimiter=",", usecols=(0, 1, 2))
    v_PA, v_DA, PV = patient_data[:, 0], patient_data[:, 1], patient_data[:, 2]
    len_v = len(v_PA)
    t_v = np.arange(len_v)
    tnew = np.linspace(0, len_v, num=int(T / dt))
    interp_DA, interp_PA, interp_P = np.interp(tnew, t_v, v_DA), np.interp(tnew, t_v, v_PA), np.interp(tnew, t_v, PV)

    u_inflow_exp1 = VelInPara(t=0.0, dt=dt, vel_t_ramp=vel_t_ramp, n=normal1, dsi=dsi1, mesh=mesh,
                              interp_velocity=interp_PA, degree=v_deg)
    u_inflow_exp2 = VelInPara(t=0.0, dt=dt, vel_t_ramp=vel_t_ramp, n=normal2, dsi=dsi2, mesh=mesh,
                              interp_velocity=interp_DA, degree=v_deg)
    u_inlet1 = DirichletBC(DVP.sub(1), u_inflow_exp1, boundaries, inlet_id1)
    u_inlet2 = DirichletBC(DVP.sub(1), u_inflow_exp2, boundaries, inlet_id2)
    u_inlet_s1 = DirichletBC(DVP.sub(1), (0.0, 0.0, 0.0), boundaries, rigid_id[0])
    u_inlet_s2 = DirichletBC(DVP.sub(1), (0.0, 0.0, 0.0), boundaries, rigid_id[1])
    d_inlet1 = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_id1)
    d_inlet2 = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_id2)
    d_inlet_s1 = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, rigid_id[0])
    d_inlet_s2 = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, rigid_id[1])

    p_out_bc_val = InnerP(t=0.0, dt=dt, interp_P=interp_P, p_t_ramp_start=p_t_ramp_start, p_t_ramp_end=p_t_ramp_end,
                          degree=p_deg)
    F_solid_linear += SurfacePressureTerm(p_out_bc_val, boundaries, fsi_id[0])
    F_solid_linear += SurfacePressureTerm(p_out_bc_val, boundaries, fsi_id[1])

    bcs = [u_inlet1, u_inlet2, u_inlet_s1, u_inlet_s2, d_inlet1, d_inlet2, d_inlet_s1, d_inlet_s2]
    inlet_area = float(dsi1[1].sum())
    return dict(bcs=bcs, u_inflow_exp1=u_inflow_exp1, u_inflow_exp2=u_inflow_exp2, p_out_bc_val=p_out_bc_val,
                F_solid_linear=F_solid_linear, n=dsi1[2], inlet_area=inlet_area, dsi1=dsi1)


def initiate(mesh_path, scale_probe, **namespace):
    probe_points = load_probe_points(mesh_path)
    if scale_probe:
        probe_points = probe_points * 0.001
    return dict(probe_points=probe_points)


def pre_solve(t, u_inflow_exp1, u_inflow_exp2, p_out_bc_val, **namespace):
    u_inflow_exp1.update(t)
    u_inflow_exp2.update(t)
    p_out_bc_val.update(t)
    return dict(u_inflow_exp1=u_inflow_exp1, u_inflow_exp2=u_inflow_exp2, p_out_bc_val=p_out_bc_val)


def post_solve(dvp_, n, dsi1, dt, mesh, inlet_area, mu_f, rho_f, probe_points, **namespace):
    d = dvp_["n"].sub(0, deepcopy=True)
    v = dvp_["n"].sub(1, deepcopy=True)
    p = dvp_["n"].sub(2, deepcopy=True)
    print_probe_points(v, p, probe_points)
    calculate_and_print_flow_properties(dt, mesh, v, inlet_area, mu_f, rho_f, n, dsi1)
    compute_minimum_jacobian(mesh, d)
