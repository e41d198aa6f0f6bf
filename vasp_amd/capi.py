"""ctypes binding of ``libvaspfsi.so`` (``include/vaspfsi.h``) and the backend the time-step driver uses.

There is no CPU fallback: if the HIP library has not been built, or no GPU is visible, constructing a
``HipBackend`` raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import List, Optional

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "libvaspfsi.so"

FSI_OK = 0
ERROR_NAMES = {1: "FSI_ERR_INVALID", 2: "FSI_ERR_DEVICE", 3: "FSI_ERR_DIVERGED", 4: "FSI_ERR_LINEAR", 5: "FSI_ERR_PIVOT"}

EXPORTED_SYMBOLS = (
    "fsi_create", "fsi_destroy", "fsi_last_error", "fsi_set_dirichlet", "fsi_set_dirichlet_values",
    "fsi_set_pressure_facets", "fsi_set_interface_pressure", "fsi_set_robin_facets", "fsi_solver_setup",
    "fsi_assemble_residual", "fsi_assemble_jacobian", "fsi_solve", "fsi_newton_solve", "fsi_shift",
    "fsi_get_state", "fsi_set_state", "fsi_num_dofs", "fsi_matrix_nnz", "fsi_device_memory", "fsi_apply_preconditioner", "fsi_get_matrix", "fsi_spmv",
    "fsi_get_timers", "fsi_get_solver_events", "fsi_xcd_order", "fsi_bcr_plan_graph", "fsi_solid_coarse_info", "fsi_solid_coarse_matrix", "fsi_solid_coarse_solve", "fsi_get_values", "fsi_stress_strain", "fsi_wall_shear_stress", "fsi_calibration_streams", "fsi_set_newton_forcing", "fsi_set_linear_solver", "fsi_set_chebyshev", "fsi_probe", "fsi_flow_stats", "fsi_set_partition",
    "fsi_rccl_unique_id", "fsi_set_rccl", "fsi_create_tuned", "fsi_get_tuning", "fsi_tuning_defaults", "fsi_tuning_from_env", "fsi_tuning_copy_out",
)


class FsiMeshDesc(C.Structure):
    _fields_ = [("num_vertices", C.c_int64), ("num_nodes", C.c_int64), ("num_cells", C.c_int64),
                ("coords", C.c_void_p), ("tet_nodes", C.c_void_p), ("cell_kind", C.c_void_p),
                ("cell_region", C.c_void_p)]


class FsiParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("theta", C.c_double), ("num_fluid_regions", C.c_int32),
                ("fluid_props", C.c_void_p), ("num_solid_regions", C.c_int32), ("solid_props", C.c_void_p),
                ("solid_models", C.c_void_p), ("delta", C.c_double), ("laplace_alpha", C.c_double)]


class FsiTuning(C.Structure):
    """include/vaspfsi.h: every product option of a context (storage precisions, sizes, Newton / Krylov policy, the
    preconditioner's structure and sweep counts).  ``HipBackend(desc, tuning={"krylov_fp32": 0, ...})`` starts from the
    library's defaults with the environment's FSI_<NAME> overrides and sets the named fields."""
    _fields_ = [("struct_size", C.c_int32), ("krylov_fp32", C.c_int32), ("operator_fp32", C.c_int32), ("schur_fp32", C.c_int32),
                ("sweeps_fp32", C.c_int32), ("sweeps_fp16", C.c_int32), ("solid_fp32", C.c_int32), ("pv_fp32", C.c_int32),
                ("krylov_capacity", C.c_int32), ("krylov_fp32_floor", C.c_double),
                ("assembly_atomic", C.c_int32), ("node_order", C.c_int32), ("tiles", C.c_int32), ("tile_nodes", C.c_int32), ("schur_tile_rows", C.c_int32), ("jacobian_waves", C.c_int32),
                ("jacobian_mfma", C.c_int32),
                ("newton_forcing", C.c_double), ("newton_forcing_late", C.c_double), ("newton_late_factor", C.c_double),
                ("f32_cycle_floor", C.c_double), ("f32_verdict_skip_rtol", C.c_double), ("orth_floor32", C.c_double),
                ("orth_floor64", C.c_double), ("gcr_escape", C.c_double), ("gcr_reorth", C.c_double),
                ("prec_streams", C.c_int32), ("experiment", C.c_int32), ("cheb4", C.c_int32), ("coarse_power", C.c_int32), ("solid_mg", C.c_int32),
                ("dd_mg", C.c_int32), ("mg_keep", C.c_int32), ("solid_block_jacobi", C.c_int32), ("solid_fused", C.c_int32),
                ("fused_sweeps", C.c_int32), ("scalar_dd", C.c_int32),
                ("its_solid", C.c_int32), ("its_fluid", C.c_int32), ("its_schur", C.c_int32), ("its_disp", C.c_int32),
                ("kappa_solid", C.c_double), ("kappa_fluid", C.c_double), ("kappa_schur", C.c_double), ("kappa_disp", C.c_double),
                ("sbmg_pre", C.c_int32), ("sbmg_post", C.c_int32), ("sbmg_cits", C.c_int32), ("sbmg_alpha", C.c_double),
                ("sbmg_ckappa", C.c_double),
                ("mg_pre", C.c_int32), ("mg_post", C.c_int32), ("mg_cits", C.c_int32), ("mg_alpha", C.c_double), ("mg_ckappa", C.c_double),
                ("solid_coarse_exact", C.c_int32), ("compact_drows", C.c_int32), ("bcr_shift", C.c_double), ("newton_adaptive", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class FsiNewtonOpts(C.Structure):
    _fields_ = [("atol", C.c_double), ("rtol", C.c_double), ("max_it", C.c_int32), ("lmbda", C.c_double),
                ("recompute", C.c_int32), ("recompute_tstep", C.c_int32), ("counter", C.c_int32),
                ("first_step_num", C.c_int32), ("lin_rtol", C.c_double), ("lin_max_it", C.c_int32),
                ("lin_solver", C.c_int32)]


class FsiNewtonIter(C.Structure):
    _fields_ = [("residual", C.c_double), ("rel_res", C.c_double), ("recomputed", C.c_int32),
                ("lin_iters", C.c_int32), ("lin_relres", C.c_double)]


class FsiTimers(C.Structure):
    _fields_ = [("residual_ms", C.c_double), ("residual_calls", C.c_int64), ("jacobian_ms", C.c_double),
                ("jacobian_calls", C.c_int64), ("factor_ms", C.c_double), ("factor_calls", C.c_int64),
                ("spmv_ms", C.c_double), ("spmv_calls", C.c_int64), ("precond_ms", C.c_double),
                ("precond_calls", C.c_int64), ("ortho_ms", C.c_double), ("ortho_calls", C.c_int64),
                ("krylov_ms", C.c_double), ("krylov_solves", C.c_int64), ("krylov_iters", C.c_int64),
                ("inner_vv_iters", C.c_int64), ("inner_schur_iters", C.c_int64), ("inner_dd_iters", C.c_int64),
                ("precond_applies", C.c_int64), ("solid_spmv_ms", C.c_double), ("solid_spmv_calls", C.c_int64),
                ("solid_nnz", C.c_int64), ("solid_rows", C.c_int64), ("db_spmv_ms", C.c_double),
                ("db_spmv_calls", C.c_int64), ("db_pairs", C.c_int64), ("db_nodes", C.c_int64),
                ("sc_spmv_ms", C.c_double), ("sc_spmv_calls", C.c_int64), ("disp_scalar", C.c_int64), ("tile_entries", C.c_int64),
                ("ortho_q_cols", C.c_int64), ("ortho_q_launches", C.c_int64), ("ortho_z_cols", C.c_int64),
                ("ortho_z_launches", C.c_int64), ("q_elem_bytes", C.c_int64), ("ldq", C.c_int64), ("ldz", C.c_int64),
                ("krylov_dirs", C.c_int64), ("krylov_cap", C.c_int64), ("schur_nnz", C.c_int64), ("schur_rows", C.c_int64),
                ("flush_ms", C.c_double), ("flush_calls", C.c_int64), ("schur_ms", C.c_double), ("schur_calls", C.c_int64),
                ("schur_elem_bytes", C.c_int64), ("node_pairs", C.c_int64),
                ("node_vertex_pairs", C.c_int64), ("spmv_fp32_calls", C.c_int64), ("sweep_flags", C.c_int64), ("part_allreduces", C.c_int64),
                ("assembly_colours", C.c_int64), ("gcr_arnoldi_steps", C.c_int64), ("gcr_restarts", C.c_int64),
                ("newton_retries", C.c_int64), ("fp32_fallbacks", C.c_int64), ("verdicts_skipped", C.c_int64),
                ("reorth_forced", C.c_int64), ("dd_cache_hits", C.c_int64), ("newton_late_solves", C.c_int64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class FsiError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


_lib = None


def load_library(path: Optional[Path] = None):
    """Load libvaspfsi.so and declare the prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise RuntimeError(f"{p} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
                           f"or make -C vasp_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(str(p))
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    lib.fsi_create.argtypes = [C.POINTER(FsiMeshDesc), C.POINTER(FsiParams), C.c_int, C.POINTER(vp)]
    lib.fsi_create_tuned.argtypes = [C.POINTER(FsiMeshDesc), C.POINTER(FsiParams), C.c_int, C.POINTER(FsiTuning), C.POINTER(vp)]
    lib.fsi_get_tuning.argtypes = [vp, C.POINTER(FsiTuning)]
    lib.fsi_tuning_defaults.argtypes = [C.POINTER(FsiTuning)]
    lib.fsi_tuning_from_env.argtypes = [C.POINTER(FsiTuning)]
    lib.fsi_tuning_copy_out.argtypes = [C.POINTER(FsiTuning), C.POINTER(FsiTuning)]
    lib.fsi_destroy.argtypes = [vp]
    lib.fsi_last_error.argtypes = [vp]
    lib.fsi_last_error.restype = C.c_char_p
    lib.fsi_set_dirichlet.argtypes = [vp, i64, vp]
    lib.fsi_set_dirichlet_values.argtypes = [vp, i64, vp]
    lib.fsi_set_pressure_facets.argtypes = [vp, i64, vp, vp]
    lib.fsi_set_interface_pressure.argtypes = [vp, dbl]
    lib.fsi_set_robin_facets.argtypes = [vp, i64, vp, vp, vp]
    lib.fsi_solver_setup.argtypes = [vp]
    lib.fsi_assemble_residual.argtypes = [vp, C.POINTER(dbl)]
    lib.fsi_assemble_jacobian.argtypes = [vp]
    lib.fsi_solve.argtypes = [vp, dbl, i32, i32, C.POINTER(i32), C.POINTER(dbl)]
    lib.fsi_newton_solve.argtypes = [vp, C.POINTER(FsiNewtonOpts), C.POINTER(FsiNewtonIter), C.POINTER(i32)]
    lib.fsi_shift.argtypes = [vp]
    lib.fsi_get_state.argtypes = [vp, C.c_int, vp]
    lib.fsi_set_state.argtypes = [vp, C.c_int, vp]
    lib.fsi_get_values.argtypes = [vp, C.c_int, i64, vp, vp]
    lib.fsi_calibration_streams.argtypes = [vp, i64]
    lib.fsi_set_newton_forcing.argtypes = [vp, dbl]
    lib.fsi_stress_strain.argtypes = [vp, i64, vp, vp]
    lib.fsi_wall_shear_stress.argtypes = [vp, i64, vp, vp, dbl, vp]
    lib.fsi_num_dofs.argtypes = [vp]
    lib.fsi_num_dofs.restype = i64
    lib.fsi_matrix_nnz.argtypes = [vp]
    lib.fsi_matrix_nnz.restype = i64
    lib.fsi_device_memory.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.fsi_get_matrix.argtypes = [vp, vp, vp, vp]
    lib.fsi_spmv.argtypes = [vp, vp, vp]
    lib.fsi_apply_preconditioner.argtypes = [vp, vp, vp]
    lib.fsi_get_timers.argtypes = [vp, C.POINTER(FsiTimers), C.c_int]
    lib.fsi_get_solver_events.argtypes = [vp, vp]
    lib.fsi_bcr_plan_graph.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.fsi_xcd_order.argtypes = [i64, vp]
    lib.fsi_xcd_order.restype = i64
    lib.fsi_solid_coarse_info.argtypes = [vp, vp]
    lib.fsi_solid_coarse_matrix.argtypes = [vp, vp, vp, vp]
    lib.fsi_solid_coarse_solve.argtypes = [vp, vp, vp]
    lib.fsi_set_linear_solver.argtypes = [vp, i32]
    lib.fsi_probe.argtypes = [vp, i64, vp, vp, vp]
    lib.fsi_flow_stats.argtypes = [vp, vp]
    lib.fsi_set_chebyshev.argtypes = [vp, i32, dbl, i32, dbl, i32, dbl, i32, dbl]
    lib.fsi_set_partition.argtypes = [vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, vp]
    lib.fsi_rccl_unique_id.argtypes = [vp]
    lib.fsi_set_rccl.argtypes = [vp, C.c_char_p, i32, i32, vp, vp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name in ("fsi_tuning_defaults", "fsi_tuning_from_env", "fsi_tuning_copy_out"):
            fn.restype = None
        elif name not in ("fsi_last_error", "fsi_num_dofs", "fsi_matrix_nnz"):
            fn.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


STATE = {"n": 0, "n-1": 1, "b": 2, "du": 3}


def _spread3(x: np.ndarray) -> np.ndarray:
    """Bits of a 21-bit integer spread to every third position (for a 63-bit Morton key)."""
    x = x.astype(np.uint64) & np.uint64(0x1FFFFF)
    x = (x | (x << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    x = (x | (x << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    x = (x | (x << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    x = (x | (x << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
    return x


def cell_locality_order(coords: np.ndarray, tets: np.ndarray, num_owned: Optional[int] = None) -> np.ndarray:
    """Permutation that puts the cells along a Morton (Z-order) curve through their centroids - the curve the library
    numbers the nodes along - so that the ~24 cells around a node are assembled close in time and the node's state, its
    residual entries and its matrix rows are still on die when the next of them needs them.  A mesh file's own order need
    not have that property: the generator of the bench mesh emits twelve sweeps over the whole domain (one tetrahedron per
    hexahedron and sweep), which made every element kernel fetch each node twelve times (k_residual: 3.4 GB of HBM traffic
    for 1.9 GB of algorithmic bytes).  With ``num_owned`` (element partition: owned cells first, then ghost cells) the two
    groups are ordered separately."""
    c = np.asarray(coords, dtype=np.float64)[np.asarray(tets)[:, :4]].mean(axis=1)
    lo, ext = c.min(axis=0), np.maximum(np.ptp(c, axis=0), 1e-300)
    q = np.minimum(((c - lo) / ext.max() * (2 ** 21 - 1)).astype(np.int64), 2 ** 21 - 1)
    key = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << np.uint64(1)) | (_spread3(q[:, 2]) << np.uint64(2))
    if num_owned is None:
        return np.argsort(key, kind="stable")
    no = int(num_owned)
    return np.concatenate([np.argsort(key[:no], kind="stable"), no + np.argsort(key[no:], kind="stable")])


class HipBackend:
    """One problem instance resident on one GPU; the methods are what ``monolithic.run`` calls per time step."""

    def __init__(self, desc: dict, device: int = 0, lin_rtol: float = 1e-10, lin_max_it: int = 4000,
                 lin_solver: int = 0, precond: int = 0,
                 newton_forcing: Optional[float] = None, num_owned_cells: Optional[int] = None,
                 tuning: Optional[dict] = None):
        import os
        self.lib = load_library()
        self.ctx = C.c_void_p()
        self.lin_rtol, self.lin_max_it, self.lin_solver = lin_rtol, lin_max_it, lin_solver
        coords = np.ascontiguousarray(desc["coords"], dtype=np.float64)
        # cells are handed to the library in a locality order (VASPFSI_CELL_ORDER=mesh: as the caller numbers them); cell
        # indices at this boundary (probes, stress cells, facet cells) stay the caller's and are mapped here
        tn_user = np.asarray(desc["tet_nodes"])
        if os.environ.get("VASPFSI_CELL_ORDER", "morton") == "mesh":
            order = np.arange(len(tn_user))
        else:
            order = cell_locality_order(coords, tn_user, num_owned_cells)
        self.cell_order = order
        self.cell_u2i = np.empty(len(order), dtype=np.int32)
        self.cell_u2i[order] = np.arange(len(order), dtype=np.int32)
        tet_nodes = np.ascontiguousarray(tn_user[order], dtype=np.int32)
        kind = np.ascontiguousarray(np.asarray(desc["cell_kind"])[order], dtype=np.int32)
        region = np.ascontiguousarray(np.asarray(desc["cell_region"])[order], dtype=np.int32)
        fprops = np.ascontiguousarray(np.asarray(desc["fluid_props"], dtype=np.float64).reshape(-1, 2))
        sp_rows = [tuple(r) + (0.0,) * (6 - len(r)) for r in desc["solid_props"]]     # rho, mu, lambda[, C10, C01, C11]
        sprops = np.ascontiguousarray(np.asarray(sp_rows, dtype=np.float64).reshape(-1, 6))
        smodels = np.ascontiguousarray(desc.get("solid_models", [0] * len(sprops)), dtype=np.int32)
        md = FsiMeshDesc(len(coords), int(desc["num_nodes"]), len(tet_nodes), _ptr(coords), _ptr(tet_nodes),
                         _ptr(kind), _ptr(region))
        pr = FsiParams(float(desc["dt"]), float(desc["theta"]), len(fprops), _ptr(fprops), len(sprops), _ptr(sprops),
                       _ptr(smodels), float(desc.get("delta", 1.0e7)), float(desc.get("laplace_alpha", 1.0)))
        if tuning:          # named FsiTuning fields over the defaults + environment overrides; fsi_create_tuned takes the struct as given
            t = FsiTuning()
            self.lib.fsi_tuning_from_env(C.byref(t))
            for k, v in tuning.items():
                if k not in dict(FsiTuning._fields_):
                    raise KeyError(f"FsiTuning has no field {k!r}")
                setattr(t, k, v)
            rc = self.lib.fsi_create_tuned(C.byref(md), C.byref(pr), device, C.byref(t), C.byref(self.ctx))
        else:
            rc = self.lib.fsi_create(C.byref(md), C.byref(pr), device, C.byref(self.ctx))
        if rc != FSI_OK:
            msg = self.lib.fsi_last_error(self.ctx).decode() if self.ctx else "fsi_create failed"
            if self.ctx:
                self.lib.fsi_destroy(self.ctx)
                self.ctx = C.c_void_p()
            raise FsiError(rc, msg)
        self.ndof = int(self.lib.fsi_num_dofs(self.ctx))
        self._check(self.lib.fsi_set_linear_solver(self.ctx, int(precond)))
        if newton_forcing is not None:          # 0: every Newton system solved to lin_rtol, as the reference's direct LU
            self._check(self.lib.fsi_set_newton_forcing(self.ctx, float(newton_forcing)))
        bc = np.ascontiguousarray(desc.get("bc_dofs", np.zeros(0)), dtype=np.int64)
        self._check(self.lib.fsi_set_dirichlet(self.ctx, len(bc), _ptr(bc)))
        self.nbc = len(bc)
        pf = desc.get("pressure_facets")
        if pf is not None and len(pf):
            pf = np.ascontiguousarray(pf, dtype=np.int32)
            pc = np.ascontiguousarray(self.cell_u2i[np.asarray(desc["pressure_facet_cell"], dtype=np.int64)], dtype=np.int32)
            self._check(self.lib.fsi_set_pressure_facets(self.ctx, len(pf), _ptr(pf), _ptr(pc)))
        rf = desc.get("robin_facets")
        if rf is not None and len(rf):
            rf = np.ascontiguousarray(rf, dtype=np.int32)
            rk = np.ascontiguousarray(desc["robin_k"], dtype=np.float64)
            rcoef = np.ascontiguousarray(desc["robin_c"], dtype=np.float64)
            self._check(self.lib.fsi_set_robin_facets(self.ctx, len(rf), _ptr(rf), _ptr(rk), _ptr(rcoef)))
        self._check(self.lib.fsi_solver_setup(self.ctx))
        self.history: List[list] = []
        self._flow_stats = None

    # ---- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != FSI_OK:
            raise FsiError(rc, self.lib.fsi_last_error(self.ctx).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.fsi_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- backend protocol of monolithic.run -----------------------------------------------------------
    def set_dirichlet_values(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self.lib.fsi_set_dirichlet_values(self.ctx, len(v), _ptr(v)))

    def set_interface_pressure(self, P: float):
        self._check(self.lib.fsi_set_interface_pressure(self.ctx, float(P)))

    def newton_solve(self, *, counter, first_step_num, atol, rtol, max_it, lmbda, recompute, recompute_tstep, log=None):
        opts = FsiNewtonOpts(atol, rtol, int(max_it), lmbda, int(recompute), int(recompute_tstep), int(counter),
                             int(first_step_num), self.lin_rtol, self.lin_max_it, self.lin_solver)
        iters = (FsiNewtonIter * int(max_it))()
        n = C.c_int32(0)
        self._flow_stats = None
        rc = self.lib.fsi_newton_solve(self.ctx, C.byref(opts), iters, C.byref(n))
        hist = []
        for i in range(n.value):
            it = iters[i]
            if log:
                if it.recomputed:
                    log("Compute Jacobian matrix")
                log("Newton iteration %d: r (atol) = %.3e (tol = %.3e), r (rel) = %.3e (tol = %.3e) "
                    % (i, it.residual, atol, it.rel_res, rtol))
            hist.append((it.residual, it.rel_res, bool(it.recomputed), it.lin_iters, it.lin_relres))
        self.history.append(hist)
        if rc == 3:
            raise RuntimeError("Error: The simulation has diverged during the Newton solve.")
        self._check(rc)
        return hist

    def shift(self):
        self._check(self.lib.fsi_shift(self.ctx))

    def get_state(self, which, out=None):
        out = np.empty(self.ndof) if out is None else out
        self._check(self.lib.fsi_get_state(self.ctx, STATE[which], _ptr(out)))
        return out

    def get_values(self, which, dofs):
        """state[dofs] (user layout) without copying the whole vector off the device."""
        dofs = np.ascontiguousarray(dofs, dtype=np.int64)
        out = np.empty(len(dofs))
        self._check(self.lib.fsi_get_values(self.ctx, STATE[which], len(dofs), _ptr(dofs), _ptr(out)))
        return out

    def set_state(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.ndof,)
        self._flow_stats = None
        self._check(self.lib.fsi_set_state(self.ctx, STATE[which], _ptr(x)))

    # ---- pieces of the hot path (tests, benchmarks) ----------------------------------------------------
    def assemble_residual(self) -> float:
        nrm = C.c_double(0.0)
        self._check(self.lib.fsi_assemble_residual(self.ctx, C.byref(nrm)))
        return nrm.value

    def assemble_jacobian(self):
        self._check(self.lib.fsi_assemble_jacobian(self.ctx))

    def solve(self, lin_rtol=None, lin_max_it=None, lin_solver=None):
        it, rr = C.c_int32(0), C.c_double(0.0)
        self._check(self.lib.fsi_solve(self.ctx, self.lin_rtol if lin_rtol is None else lin_rtol,
                                       self.lin_max_it if lin_max_it is None else lin_max_it,
                                       self.lin_solver if lin_solver is None else lin_solver, C.byref(it), C.byref(rr)))
        return it.value, rr.value

    def device_memory(self):
        """(free, total) bytes of the context's device, from the library's own HIP runtime."""
        f, t = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.fsi_device_memory(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def matrix(self):
        """The assembled Jacobian (after ident_zeros and bc.apply) as scipy CSR in the user dof layout."""
        import scipy.sparse as sp
        nnz = int(self.lib.fsi_matrix_nnz(self.ctx))
        rp = np.empty(self.ndof + 1, dtype=np.int64)
        ci = np.empty(nnz, dtype=np.int64)
        va = np.empty(nnz, dtype=np.float64)
        self._check(self.lib.fsi_get_matrix(self.ctx, _ptr(rp), _ptr(ci), _ptr(va)))
        return sp.csr_matrix((va, ci, rp), shape=(self.ndof, self.ndof))

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.ndof)
        self._check(self.lib.fsi_spmv(self.ctx, _ptr(x), _ptr(y)))
        return y

    def apply_preconditioner(self, r):
        """z = M^-1 r: one application of the active preconditioner (user layout), as the Krylov method applies it."""
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.empty(self.ndof)
        self._check(self.lib.fsi_apply_preconditioner(self.ctx, _ptr(r), _ptr(z)))
        return z

    def set_linear_solver(self, precond: int = 0):
        self._check(self.lib.fsi_set_linear_solver(self.ctx, int(precond)))

    def set_chebyshev(self, its_solid=0, kappa_solid=0.0, its_fluid=0, kappa_fluid=0.0, its_schur=0, kappa_schur=0.0,
                      its_disp=0, kappa_disp=0.0):
        self._check(self.lib.fsi_set_chebyshev(self.ctx, int(its_solid), float(kappa_solid), int(its_fluid),
                                               float(kappa_fluid), int(its_schur), float(kappa_schur), int(its_disp),
                                               float(kappa_disp)))

    def probe(self, cells, bary):
        """(n,7) array d(3) v(3) p of dvp_["n"] at located points (cells (n,), barycentric (n,4))."""
        cells = np.ascontiguousarray(self.cell_u2i[np.asarray(cells, dtype=np.int64)], dtype=np.int32)
        bary = np.ascontiguousarray(bary, dtype=np.float64)
        out = np.empty((len(cells), 7))
        self._check(self.lib.fsi_probe(self.ctx, len(cells), _ptr(cells), _ptr(bary), _ptr(out)))
        return out

    def flow_stats(self):
        """(mean, min, max of the cell-mean |v|, min of the cell-mean det(I + grad d)) of dvp_["n"].  post_solve asks twice
        per step (flow properties, minimum Jacobian [REF src/vasp/simulations/simulation_common.py:253-348]): the second call of
        a step returns the first one's numbers (the state changes only through newton_solve / set_state)."""
        if self._flow_stats is None:
            out = np.empty(4)
            self._check(self.lib.fsi_flow_stats(self.ctx, _ptr(out)))
            self._flow_stats = tuple(out)
        return self._flow_stats

    def stress_strain(self, cells):
        """DG1 Cauchy stress / Green-Lagrange strain / largest principal values on solid ``cells`` of dvp_["n"]."""
        cells = np.ascontiguousarray(self.cell_u2i[np.asarray(cells, dtype=np.int64)], dtype=np.int32)
        out = np.empty((len(cells), 80))
        self._check(self.lib.fsi_stress_strain(self.ctx, len(cells), _ptr(cells), _ptr(out)))
        return dict(TrueStress=out[:, :36].reshape(-1, 4, 3, 3), GreenLagrangeStrain=out[:, 36:72].reshape(-1, 4, 3, 3),
                    MaxPrincipalStress=out[:, 72:76].copy(), MaxPrincipalStrain=out[:, 76:80].copy())

    def wall_shear_stress(self, facet_cells, facet_local, mu: float):
        """(nf, 3, 3) projected tangential traction at the vertices of exterior facets (cell, opposite local vertex)."""
        fc = np.ascontiguousarray(self.cell_u2i[np.asarray(facet_cells, dtype=np.int64)], dtype=np.int32)
        fl = np.ascontiguousarray(facet_local, dtype=np.int32)
        out = np.empty((len(fc), 3, 3))
        self._check(self.lib.fsi_wall_shear_stress(self.ctx, len(fc), _ptr(fc), _ptr(fl), float(mu), _ptr(out)))
        return out

    def tuning(self) -> dict:
        """The FsiTuning the context was created with."""
        t = FsiTuning()
        self._check(self.lib.fsi_get_tuning(self.ctx, C.byref(t)))
        return t.as_dict()

    # ---- the solid cycle's coarse level (test hooks of the exact solve, csrc/fsi_bcr.hip) ---------------------------------
    def solid_coarse_info(self) -> dict:
        out = np.zeros(12, dtype=np.int64)
        self._check(self.lib.fsi_solid_coarse_info(self.ctx, _ptr(out)))
        keys = ("nodes", "blocks3x3", "planned", "ready", "bfs_blocks", "levels", "operator_bytes", "launches_per_solve", "max_block",
                "solves", "setup_flops", "cycle_ready")
        return {k: int(v) for k, v in zip(keys, out)}

    def solid_coarse_matrix(self):
        """The coarse operator as scipy CSR (3 unknowns per coarse node)."""
        import scipy.sparse as sp
        info = self.solid_coarse_info()
        nc, nb = info["nodes"], info["blocks3x3"]
        cptr, ccol, cvals = np.empty(nc + 1, dtype=np.int64), np.empty(nb, dtype=np.int32), np.empty(9 * nb, dtype=np.float32)
        self._check(self.lib.fsi_solid_coarse_matrix(self.ctx, _ptr(cptr), _ptr(ccol), _ptr(cvals)))
        return sp.bsr_matrix((cvals.reshape(nb, 3, 3).astype(np.float64), ccol, cptr), shape=(3 * nc, 3 * nc)).tocsr(), cptr, ccol

    def solid_coarse_solve(self, rhs):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.empty_like(rhs)
        self._check(self.lib.fsi_solid_coarse_solve(self.ctx, _ptr(rhs), _ptr(x)))
        return x

    def solver_events(self) -> dict:
        """Run totals (since the context was created; timer resets do not touch them) of what the linear solver had to do beyond
        iterating - a cheap read: no device synchronisation, unlike ``timers()``."""
        out = np.zeros(8, dtype=np.int64)
        self._check(self.lib.fsi_get_solver_events(self.ctx, _ptr(out)))
        return {"newton_retries": int(out[0]), "fp32_fallbacks": int(out[1]), "gcr_restarts": int(out[2]),
                "adaptive_solves": int(out[3]), "adaptive_tightened": int(out[4]), "exact_coarse_solves": int(out[5])}

    def timers(self, reset=False) -> dict:
        t = FsiTimers()
        self._check(self.lib.fsi_get_timers(self.ctx, C.byref(t), int(reset)))
        return t.as_dict()
