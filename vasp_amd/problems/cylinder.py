"""Tiny-cylinder FSI problem, re-hosted without DOLFIN.

Same parameters, boundary conditions and hooks as [REF src/vasp/simulations/cylinder.py]: parabolic inlet
velocity with a cosine ramp (:89-130), constant ramped wall pressure on the FSI interface (:133-167),
zero-velocity / zero-displacement sets (:180-189).  Smallest case; most of the reference's tests run it.
"""
import numpy as np

from vasp_amd.problems import *  # noqa: F401,F403
from vasp_amd.problems import _compiler_parameters
from vasp_amd.fem import DirichletBC, SurfacePressureTerm
from vasp_amd.mesh import FsiMesh
from vasp_amd.simulation_common import calculate_and_print_flow_properties


def set_problem_parameters(default_variables, **namespace):
    E_s_val, nu_s_val = 1e6, 0.45
    mu_s_val = E_s_val / (2 * (1 + nu_s_val))
    lambda_s_val = nu_s_val * 2.0 * mu_s_val / (1.0 - 2.0 * nu_s_val)

    default_variables.update(dict(
        T=0.1, dt=0.001, theta=0.501, save_step=1, checkpoint_step=50,
        linear_solver="mumps", atol=1e-6, rtol=1e-6, recompute=20, recompute_tstep=20,
        mesh_path="mesh/cylinder.h5",
        inlet_id=2, inlet_outlet_s_id=11, fsi_id=22, rigid_id=11, outer_wall_id=33,
        rho_f=1.025e3, mu_f=3.5e-3, dx_f_id=1, v_max_final=0.75, P_final=10000,
        rho_s=1.0e3, mu_s=mu_s_val, nu_s=nu_s_val, lambda_s=lambda_s_val, dx_s_id=2,
        extrapolation="laplace", extrapolation_sub_type="constant",
        folder="cylinder_results", save_deg=1,
    ))
    return default_variables


def get_mesh_domain_and_boundaries(mesh_path, **namespace):
    print("Obtaining mesh, domains and boundaries...")
    mesh = FsiMesh.read(mesh_path)
    return mesh, mesh.cell_markers, mesh.facet_markers


class VelInPara:
    """Parabolic inlet profile -n v(t) (1 - r^2/R^2), R = sqrt(A/pi), r measured from the patch barycentre."""

    def __init__(self, t, t_ramp, v_max_final, n, dsi, mesh, **kwargs):
        self.t, self.t_ramp, self.v_max_final, self.v, self.n = t, t_ramp, v_max_final, 0.0, np.asarray(n)
        fids, area, _ = dsi
        self.A = float(area.sum())
        self.c = (area[:, None] * mesh.coords[mesh.facets[fids]].mean(axis=1)).sum(axis=0) / self.A
        self.r = np.sqrt(self.A / np.pi)

    def update(self, t):
        self.t = t
        ramp_factor = -0.5 * np.cos(np.pi * t / self.t_ramp) + 0.5 if t < self.t_ramp else 1.0
        self.v = ramp_factor * self.v_max_final
        print("v (centerline, at inlet) = {} m/s".format(self.v))

    def eval_nodes(self, x):
        fact_r = 1.0 - ((x - self.c) ** 2).sum(axis=1) / self.r ** 2
        return -self.n[None, :] * self.v * fact_r[:, None]


class InnerP:
    def __init__(self, t, t_ramp, P_final, **kwargs):
        self.t, self.t_ramp, self.P_final, self.P = t, t_ramp, P_final, 0.0

    def update(self, t):
        self.t = t
        ramp_factor = -0.5 * np.cos(np.pi * t / self.t_ramp) + 0.5 if t < self.t_ramp else 1.0
        self.P = ramp_factor * self.P_final
        print("P = {} Pa".format(self.P))


def create_bcs(DVP, mesh, boundaries, P_final, v_max_final, fsi_id, inlet_id, inlet_outlet_s_id, rigid_id, psi,
               F_solid_linear, **namespace):
    # wall pressure on the interface, reference configuration [REF cylinder.py:161-167]
    p_out_bc_val = InnerP(t=0.0, t_ramp=0.1, P_final=P_final, degree=2)
    F_solid_linear += SurfacePressureTerm(p_out_bc_val, boundaries, fsi_id)

    fids = np.nonzero(boundaries == inlet_id)[0]
    area, n = mesh.facet_area_normals(fids)
    dsi = (fids, area, n)
    ni = (area[:, None] * n).sum(axis=0)
    normal = ni / np.sqrt((ni ** 2).sum())

    u_inflow_exp = VelInPara(t=0.0, t_ramp=0.1, v_max_final=v_max_final, n=normal, dsi=dsi, mesh=mesh, degree=3)
    u_inlet = DirichletBC(DVP.sub(1), u_inflow_exp, boundaries, inlet_id)
    u_inlet_s = DirichletBC(DVP.sub(1), (0.0, 0.0, 0.0), boundaries, inlet_outlet_s_id)
    d_inlet = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_id)
    d_inlet_s = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, inlet_outlet_s_id)
    d_rigid = DirichletBC(DVP.sub(0), (0.0, 0.0, 0.0), boundaries, rigid_id)
    bcs = [u_inlet, d_inlet, u_inlet_s, d_inlet_s, d_rigid]

    inlet_area = float(area.sum())
    return dict(bcs=bcs, u_inflow_exp=u_inflow_exp, p_out_bc_val=p_out_bc_val, F_solid_linear=F_solid_linear,
                dsi=dsi, inlet_area=inlet_area, n=n)


def pre_solve(t, u_inflow_exp, p_out_bc_val, **namespace):
    u_inflow_exp.update(t)
    p_out_bc_val.update(t)
    return dict(u_inflow_exp=u_inflow_exp, p_out_bc_val=p_out_bc_val)


def post_solve(dvp_, dt, mesh, inlet_area, mu_f, rho_f, n, dsi, **namespace):
    v = dvp_["n"].sub(1, deepcopy=True)
    calculate_and_print_flow_properties(dt, mesh, v, inlet_area, mu_f, rho_f, n, dsi)
