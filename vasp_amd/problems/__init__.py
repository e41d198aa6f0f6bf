"""Default parameters and no-op hooks of the problem-file API (``from turtleFSI.problems import *``).

The five VaSP problem files start from turtleFSI's ``default_variables`` and override a subset
[REF src/vasp/simulations/offset_stenosis.py:9,27-82].  Key set and order follow the dump the reference
writes to ``Checkpoint/default_variables.json``
[REF tests/test_data/hemodynamics_data/Checkpoint/default_variables.json]; values are turtleFSI's
defaults.
"""
from __future__ import annotations

_compiler_parameters = dict(quadrature_degree=6, optimize=True, representation="auto",
                            cpp_optimize=True, cpp_optimize_flags="-O2")

default_variables = dict(
    # Temporal settings
    dt=0.001, theta=0.501, T=1, t=0, counter=0,
    # Spatial settings / element degrees
    v_deg=2, p_deg=1, d_deg=2,
    # Domain markers
    dx_f_id=1, dx_s_id=2, ds_s_id=None,
    # Fluid
    fluid_properties=[], rho_f=1.0e3, mu_f=1.0,
    # Solid
    solid_properties=[], material_model="StVenantKirchoff", rho_s=1.0e3, mu_s=5.0e4, nu_s=0.45,
    lambda_s=4.5e5, k_s=0.0, c_s=0.0, gravity=None,
    # Problem setup
    fluid="fluid", solid="solid", robin_bc=False, extrapolation="laplace",
    extrapolation_sub_type="constant", bc_ids=[],
    # Solver
    linear_solver="mumps", solver="newtonsolver", atol=1e-7, rtol=1e-7, max_it=50, lmbda=1.0,
    recompute=5, recompute_tstep=1, compiler_parameters=_compiler_parameters,
    # Output
    loglevel=20, verbose=True, save_step=10, save_deg=1, checkpoint_step=20, folder="results",
    sub_folder=None, restart_folder=None, killtime=None,
)


def set_problem_parameters(default_variables, **namespace):
    return default_variables


def get_mesh_domain_and_boundaries(**namespace):
    raise NotImplementedError("a problem file must define get_mesh_domain_and_boundaries")


def initiate(**namespace):
    return {}


def create_bcs(**namespace):
    return dict(bcs=[])


def pre_solve(**namespace):
    return None


def post_solve(**namespace):
    return None


def finished(**namespace):
    return None


HOOKS = ("set_problem_parameters", "get_mesh_domain_and_boundaries", "initiate", "create_bcs",
         "pre_solve", "post_solve", "finished")
