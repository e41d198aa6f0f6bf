/*
 * vaspfsi.h — C-ABI of the MI355X-native monolithic ALE-FSI time-step kernel (libvaspfsi.so).
 *
 * Drop-in boundary.  VaSP's simulation stage hands its problem files to turtleFSI, whose
 * hot loop is, per Newton iteration, `assemble(J_nonlinear)` (+ `A_pre`, `ident_zeros`, `bc.apply(A)`),
 * `assemble(-F)`, `bc.apply(b, u)`, `LUSolver.solve`, `axpy`, `bc.apply(u)` and two norms
 * (turtleFSI/modules/newtonsolver.py `solver_setup` / `newtonsolver`; call sites in the reference:
 * src/vasp/simulations/offset_stenosis.py:9 `from turtleFSI.problems import *`, the hooks at :27,:85,:143,
 * :151,:199,:216 and the solver keys at :44-48).  Each entry point below names the reference-side call it
 * replaces.  All buffers are caller-owned host memory unless stated otherwise; every function returns an
 * `int` status (FSI_OK == 0) and never throws across the ABI.  One host thread per context; contexts are
 * independent (one per GPU).
 *
 * Global dof layout at this boundary ("user layout", vasp_amd/mesh.py): [ d: 3*N2 | v: 3*N2 | p: V ],
 * component-minor, N2 = #P2 nodes (vertices first, then edges), V = #vertices.
 */
#ifndef VASPFSI_H
#define VASPFSI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSI_OK 0
#define FSI_ERR_INVALID 1      /* bad argument / inconsistent mesh arrays                                   */
#define FSI_ERR_DEVICE 2       /* HIP runtime error (no device, out of memory, launch failure)              */
#define FSI_ERR_DIVERGED 3     /* residual or update > 1e20 or NaN: the reference raises RuntimeError here   */
#define FSI_ERR_LINEAR 4       /* Krylov breakdown / no convergence within max iterations                    */
#define FSI_ERR_PIVOT 5        /* zero or non-finite pivot in the incomplete factorisation                   */

typedef struct FsiCtx FsiCtx;

/* Mesh + discrete-problem description.  Replaces: Mesh/MeshFunction reads and
 * FunctionSpace(mesh, MixedElement([P2^3, P2^3, P1])) in turtleFSI monolithic.py, driven by
 * get_mesh_domain_and_boundaries [REF src/vasp/simulations/offset_stenosis.py:85-140]. */
typedef struct FsiMeshDesc {
  int64_t num_vertices;        /* V                                                                         */
  int64_t num_nodes;           /* N2 = V + #edges                                                           */
  int64_t num_cells;           /* C                                                                         */
  const double* coords;        /* [V][3] vertex coordinates                                                 */
  const int32_t* tet_nodes;    /* [C][10] P2 node ids, UFC local order (4 vertices, 6 edges), rows of the   */
                               /*         vertex part ascending                                             */
  const int32_t* cell_kind;    /* [C] 0 = fluid, 1 = solid.  Cell ORDER is the caller's and is kept (every cell    */
                               /* index of this API refers to it); it decides the locality of the element       */
                               /* kernels: hand the cells over along a space-filling curve (the Python binding   */
                               /* does: vasp_amd.capi.cell_locality_order), not in sweeps over the domain        */
  const int32_t* cell_region;  /* [C] index into fluid_props / solid_props                                  */
} FsiMeshDesc;

/* Material / scheme parameters.  Replaces: fluid_properties / solid_properties lists, dt, theta handed to
 * fluid_setup / solid_setup / extrapolate_setup [REF offset_stenosis.py:36-80; SURVEY.md §8a a4-a6]. */
typedef struct FsiParams {
  double dt;
  double theta;
  int32_t num_fluid_regions;
  const double* fluid_props;   /* [nf][2]  rho_f, mu_f                                                      */
  int32_t num_solid_regions;
  const double* solid_props;   /* [ns][6]  rho_s, mu_s, lambda_s, C10, C01, C11 (the last three: MooneyRivlin)      */
  const int32_t* solid_models; /* [ns]     0 = StVenantKirchoff, 1 = MooneyRivlin                           */
  double delta;                /* d_t = v penalty in the solid (turtleFSI solid.py: 1e7)                    */
  double laplace_alpha;        /* mesh-lifting coefficient ("constant": 1.0)                                */
} FsiParams;

typedef struct FsiNewtonOpts {
  double atol, rtol;           /* newtonsolver(): loop while rel_res > rtol and residual > atol             */
  int32_t max_it;
  double lmbda;                /* relaxation of the update                                                  */
  int32_t recompute;           /* re-assemble the Jacobian every `recompute` Newton iterations              */
  int32_t recompute_tstep;     /* ... and every `recompute_tstep` time steps                                */
  int32_t counter;             /* time-step counter of this call                                            */
  int32_t first_step_num;      /* counter value of the first step of this run (forces a Jacobian)           */
  double lin_rtol;             /* Krylov stop: ||r|| <= eta ||b|| (row-equilibrated norms),                  */
                               /* eta = max(lin_rtol, min(1e-2, f atol / ||b||_newton)), f = 1e-2 unless      */
                               /* fsi_set_newton_forcing changed it (f = 0: eta = lin_rtol, the direct-LU    */
                               /* policy of the reference)                               (inexact Newton)   */
  int32_t lin_max_it;          /* Krylov iteration cap per linear solve                                     */
  int32_t lin_solver;          /* 0 = GCR with recycled directions, 1 = BiCGStab                            */
} FsiNewtonOpts;

typedef struct FsiNewtonIter {
  double residual;             /* ||b||_2   ("r (atol)")                                                    */
  double rel_res;              /* ||du||_L2(Omega), dolfin.norm(Function, 'l2')  ("r (rel)")                */
  int32_t recomputed;          /* 1 if "Compute Jacobian matrix" happened before this iteration             */
  int32_t lin_iters;           /* Krylov iterations (new directions) of this iteration's linear solve       */
  double lin_relres;           /* achieved ||r||/||b||                                                      */
} FsiNewtonIter;

/* ---- tuning ---------------------------------------------------------------------------------------- */
/* Every product option of a context in one place: storage precisions, sizes, Newton / Krylov policy, the preconditioner's
 * structure and sweep counts (DESIGN.md sections 4 - 5 give the measurement behind each default).  fsi_tuning_defaults fills
 * the defaults; fsi_create takes the defaults with the developer overrides of the environment applied (FSI_<NAME> for the
 * field <name>, parsed in ONE function: csrc/fsi_tuning.hip; debugging aids FSI_DEBUG* are not options and stay environment
 * only); fsi_create_tuned takes the struct as given.  The reference has no counterpart: its linear solver is
 * `linear_solver="mumps"` [REF src/vasp/simulations/offset_stenosis.py:45]. */
typedef struct FsiTuning {
  int32_t struct_size;         /* in / out in every call that takes an FsiTuning*: set it to the caller's sizeof(FsiTuning) (0 = this
                                * header's).  fsi_tuning_defaults / _from_env / fsi_get_tuning write at most that many bytes and store
                                * the number written; fsi_create_tuned reads that many and takes the defaults for the rest. */
  /* storage precisions (all arithmetic on the Newton level and every linear-solve verdict is FP64 whatever these say) */
  int32_t krylov_fp32;         /* Krylov basis Q: 0 FP64, 1 FP32, 2 decided per Jacobian lifetime from the tolerances asked for    */
  int32_t operator_fp32;       /* 1: products inside Krylov iterations on an FP32 copy of the Jacobian values (with an FP32 basis) */
  int32_t schur_fp32;          /* 1: Schur-complement sweeps on FP16 / FP32 matrix values (vectors FP64); 0: FP64 values           */
  int32_t sweeps_fp32;         /* 1: velocity / displacement block sweeps in FP32 vectors                                          */
  int32_t sweeps_fp16;         /* 1: fine-level sweep matrices as packed FP16 records                                              */
  int32_t solid_fp32;          /* 1: solid velocity block as FP32 3x3 block-CSR                                                    */
  int32_t pv_fp32;             /* 1: FP32 node-form copies of A_vp / A_pv for the two block products of the pressure step          */
  int32_t krylov_capacity;     /* kept directions per Jacobian lifetime (bounded by half of the free HBM)                          */
  double krylov_fp32_floor;    /* krylov_fp32 == 2: FP32 basis iff the lowest linear tolerance the lifetime can be asked for is    */
                               /* above this.  1e-10 since round 4 (1e-7 before): with the columns kept orthonormal (round 3) the  */
                               /* FP32 basis + restarts from the FP64 residual reach 1e-11 on the golden runs, and the aneurysm    */
                               /* problem at its own tolerances (1e-10 / 1e-9) runs 28 % faster (profiles/r04_aneurysm_fp32.txt)   */
  /* assembly and numbering */
  int32_t assembly_atomic;     /* 0: bitwise reproducible assembly (per-dof gather, cell colours); 1: unordered atomics            */
  int32_t node_order;          /* 0: Morton curve (default), 1: the mesh's order, 2: multicolour (needed by precond = 1)           */
  int32_t tiles;               /* 1: LDS-tiled sweep kernels                                                                        */
  int32_t tile_nodes;          /* nodes per workgroup of the tiled displacement / fluid sweeps: 128 | 256; 0 = 256                     */
  int32_t schur_tile_rows;     /* rows per workgroup of the tiled Schur sweep: 64 | 128 | 256; 0 = 64 (256 in rounds 2-3)               */
  int32_t jacobian_waves;      /* waves per SIMD the refresh kernel's register budget is set for (1 | 2)                           */
  int32_t jacobian_mfma;       /* 1: element contraction of the refresh kernel on v_mfma_f64_16x16x4 (k_jacobian_mfma; measured    */
                               /*    slower in round 4: 103 against 70 ms per refresh at 1.12 M tets)                              */
  /* Newton / Krylov policy */
  double newton_forcing;       /* linear tolerance = forcing * atol / |b| (0: every system to lin_rtol)                            */
  double newton_forcing_late;  /* ... of late Newton iterations (previous update within newton_late_factor of its tolerance)       */
  double newton_late_factor;
  double f32_cycle_floor;      /* FP32 basis: a cycle reduces the residual by at most this factor before the FP64 verdict          */
  double f32_verdict_skip_rtol;/* FP32 basis, inside fsi_newton_solve: answers asked for at or above this may skip the verdict     */
  double orth_floor32, orth_floor64;   /* estimated orthogonality error of a new column that forces a second Gram-Schmidt pass     */
  double gcr_escape;           /* alpha^2 <= this |r|^2: the next direction is made from the last q instead of r                   */
  double gcr_reorth;           /* second Gram-Schmidt pass when |w'| < reorth |w| (0: automatic)                                   */
  /* block preconditioner */
  int32_t prec_streams;        /* 1: two chains of an application side by side on two HIP streams                                  */
  int32_t experiment;          /* measurement switches of the block factorisation, 0 in production: bit 0       */
                               /* pressure right-hand side from the fluid predictor only, bit 1 displacement block without velocity   */
  int32_t cheb4;               /* bit 0 / 1 / 2: 4th-kind Chebyshev sweeps in the solid cycle / displacement cycle / Schur solve   */
  int32_t coarse_power;        /* 1: coarse-level Chebyshev intervals from a power iteration (0: Gershgorin bound)                 */
  int32_t solid_mg, dd_mg;     /* two-level (P2 -> P1) cycles of the solid velocity block / the displacement block                 */
  int32_t mg_keep;             /* 1: keep the displacement block's coarse operator while three checksums say it is unchanged       */
  int32_t solid_block_jacobi, solid_fused, fused_sweeps, scalar_dd;
  int32_t its_solid, its_fluid, its_schur, its_disp;          /* sweeps (solid / disp: one-level fallbacks)                         */
  double kappa_solid, kappa_fluid, kappa_schur, kappa_disp;   /* assumed condition numbers of the Chebyshev intervals              */
  int32_t sbmg_pre, sbmg_post, sbmg_cits;                     /* solid cycle: smoothing sweeps before / after, coarse sweeps       */
  double sbmg_alpha, sbmg_ckappa;                             /* ... smoothing interval [lmax / alpha, lmax], coarse kappa         */
  int32_t mg_pre, mg_post, mg_cits;                           /* displacement cycle; mg_post 0 (default) = by context size: 7 below */
                                                              /* 1.1 M P2 nodes, 5 above (fsi_get_tuning returns what was taken)     */
  double mg_alpha, mg_ckappa;
  /* round 5 (fields are appended: struct_size tells an older caller's struct from this one) */
  int32_t solid_coarse_exact;  /* 1: the solid cycle's coarse level is solved exactly - block cyclic reduction over breadth-first levels of
                                * the solid vertices, operators refreshed with the Jacobian (csrc/fsi_bcr.hip) - instead of sbmg_cits sweeps */
  int32_t compact_drows;       /* 1: the outer product takes the three displacement rows of a node from a pair form - [dd_ii, dv_ii] per
                                * node pair, six values instead of 18 + pressure columns - extracted at every Jacobian refresh together
                                * with a check that those rows hold nothing else (they do not for the forms of SURVEY.md A.2: the mesh
                                * extension and the solid's d - v relation act per component); 0: all six value rows are streamed */
  double bcr_shift;            /* ... of A_c + bcr_shift * blockdiag(A_c): the floor below which the level's modes are damped, not inverted */
  double newton_adaptive;      /* linear tolerance >= this x the contraction the Newton iteration of the same index reached one time
                                * step ago under the same Jacobian (net of its own linear tolerance); 0: off.  fsi_newton_solve */
} FsiTuning;
void fsi_tuning_defaults(FsiTuning* t);
/* (internal helper of fsi_get_tuning, exported so that the ABI test can exercise the size rule without a device) */
void fsi_tuning_copy_out(const FsiTuning* full, FsiTuning* out);
/* the defaults with the FSI_<NAME> variables of the environment applied (what fsi_create uses) */
void fsi_tuning_from_env(FsiTuning* t);

/* ---- lifetime ------------------------------------------------------------------------------------ */
/* Replaces: space / form set-up up to solver_setup() (A_pre assembled at the zero state). `device` is the
 * HIP device ordinal. */
int fsi_create(const FsiMeshDesc* mesh, const FsiParams* params, int device, FsiCtx** out);
/* fsi_create with an explicit FsiTuning (the environment is not consulted).  tuning == NULL: the defaults. */
int fsi_create_tuned(const FsiMeshDesc* mesh, const FsiParams* params, int device, const FsiTuning* tuning, FsiCtx** out);
/* the tuning the context was created with */
int fsi_get_tuning(const FsiCtx* ctx, FsiTuning* out);
int fsi_destroy(FsiCtx* ctx);
const char* fsi_last_error(const FsiCtx* ctx);

/* ---- boundary data ------------------------------------------------------------------------------- */
/* Replaces: bcs = [DirichletBC(...), ...] [REF offset_stenosis.py:170-179].  `dofs` are unique user-layout
 * dofs with list-order precedence already resolved. */
int fsi_set_dirichlet(FsiCtx* ctx, int64_t n, const int64_t* dofs);
/* Replaces: per-step expression updates in pre_solve [REF offset_stenosis.py:199-208]; values[i] belongs to dofs[i]. */
int fsi_set_dirichlet_values(FsiCtx* ctx, int64_t n, const double* values);
/* Replaces: F_solid_linear += P * inner(n('+'), psi('+')) * dS(fsi_id) [REF offset_stenosis.py:184-190].
 * `facet_nodes` [nf][6] P2 nodes (3 vertices, 3 edges (b,c),(a,c),(a,b)); `plus_cell` [nf] the '+' cell. */
int fsi_set_pressure_facets(FsiCtx* ctx, int64_t nf, const int32_t* facet_nodes, const int32_t* plus_cell);
/* Replaces: InterfacePressure.update(t) -> P [REF src/vasp/simulations/simulation_common.py:371-395]. */
int fsi_set_interface_pressure(FsiCtx* ctx, double P);
/* Replaces: robin_bc terms of solid_setup [REF src/vasp/simulations/aneurysm.py:73-76]. */
int fsi_set_robin_facets(FsiCtx* ctx, int64_t nf, const int32_t* facet_nodes, const double* k_s, const double* c_s);
/* Inexact Newton: fsi_newton_solve asks the linear solver for max(lin_rtol, min(1e-2, forcing * atol / |b|)).  The
 * reference solves every Newton system with a direct LU (MUMPS); forcing = 0 reproduces that policy (every solve to
 * lin_rtol) for parity runs, the default 1e-2 leaves the Newton iteration counts of the bench unchanged and saves the
 * Krylov iterations that would polish an update far below the Newton tolerance. */
int fsi_set_newton_forcing(FsiCtx* ctx, double forcing);
/* Replaces: `linear_solver="mumps"` [REF offset_stenosis.py:45].  precond 0 = field-split block preconditioner (velocity /
 * pressure SIMPLE split with the solid displacement eliminated, then the displacement block; every inner solve a fixed number
 * of Jacobi-Chebyshev sweeps), 1 = multicolour ILU(0) of the monolithic matrix (needs the context created with
 * FSI_ORDER=colour).  With FsiNewtonOpts.lin_solver (0 recycled GCR, 1 BiCGStab) that makes four pairs; tested on MI355X
 * (tests/test_gpu_parity.py::test_bicgstab_and_ilu0_solve_match_sparse_lu): GCR + field split (the default) and each option
 * with the other's default partner converge to 1e-10; BiCGStab + ILU(0) together stagnates on the FSI Jacobian (a saddle
 * point with 1e7 penalty rows) and is recorded there as a strict xfail. */
int fsi_set_linear_solver(FsiCtx* ctx, int32_t precond);
/* Sweep counts and assumed condition numbers of the Chebyshev solves inside the block preconditioner (solid and
 * fluid-interior part of the velocity block, pressure Schur complement).  Non-positive arguments keep the current value.
 * Defaults (csrc/fsi_tuning.hip is the one place that holds them): solid 300 / 1e4 (one-level fallback; default: the two-level cycle,
 * 16 + 16 sweeps around the exact coarse solve), fluid 4 / 5, Schur 30 / 1e2, displacement 60 / 1e3 (one-level fallback; the default
 * displacement solve is the two-level P2 -> P1 cycle: 3 + 5 smoothing sweeps - 3 + 7 in contexts below 1.1 M nodes - around 16 coarse
 * sweeps). */
int fsi_set_chebyshev(FsiCtx* ctx, int32_t its_solid, double kappa_solid, int32_t its_fluid, double kappa_fluid,
                      int32_t its_schur, double kappa_schur, int32_t its_disp, double kappa_disp);
/* Finishes set-up: assembles A_pre = assemble(J_linear) at the current state (solver_setup()). */
int fsi_solver_setup(FsiCtx* ctx);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* b = assemble(-F); [bc.apply(b, u)]; writes ||b||_2 */
int fsi_assemble_residual(FsiCtx* ctx, double* norm);
/* A = assemble(J_nonlinear) + A_pre; ident_zeros; [bc.apply(A)]; refreshes the preconditioner */
int fsi_assemble_jacobian(FsiCtx* ctx);
/* One linear solve A du = b with the current factorisation (up_sol.solve). */
int fsi_solve(FsiCtx* ctx, double lin_rtol, int32_t lin_max_it, int32_t lin_solver, int32_t* iters, double* relres);
/* newtonsolver(): one time step.  `iters` has room for opts->max_it entries; *n_iters receives the count. */
int fsi_newton_solve(FsiCtx* ctx, const FsiNewtonOpts* opts, FsiNewtonIter* iters, int32_t* n_iters);
/* dvp_["n-1"] <- dvp_["n"] (the reference's vector shift after each step). */
int fsi_shift(FsiCtx* ctx);

/* ---- element partition across GPUs (one context per GPU) --------------------------------- */
/* Replaces: the MPI-parallel dolfin path of `mpirun -np N turtleFSI ...` (ghost updates of PETSc vectors,
 * MPI sums inside the Krylov solver and in norm(); the reference's own MPI use at this boundary:
 * src/vasp/simulations/simulation_common.py:213-220).  The context holds the cells that touch a node it owns
 * (owned cells first in FsiMeshDesc, then the ghost layer); rows of nodes owned elsewhere ("ghost" dofs) are
 * carried as identity rows and are refreshed from their owner once per Krylov iteration.  The library packs
 * `send_dofs` into `sendbuf_dev`, calls `halo_exchange` and unpacks `recvbuf_dev` into `ghost_dofs`; both buffers
 * are device memory owned by the caller (length n_send / n_ghost doubles), the callbacks are the caller's transport
 * (RCCL through torch.distributed in vasp_amd/partition.py).  Callbacks return 0 on success. */
typedef struct FsiComm {
  void* user;
  int (*allreduce_sum)(void* user, double* host_vals, int32_t n);   /* in place, over all ranks                   */
  int (*halo_exchange)(void* user);                                 /* sendbuf_dev -> the peers' recvbuf_dev       */
} FsiComm;
/* dofs are user-layout dofs of this context.  num_owned_cells: cells [0, num_owned_cells) are counted in the
 * L2(Omega) norm of the Newton update by this rank.  ghost_dofs: every dof owned elsewhere (refreshed by the halo
 * exchange, zero in residual-type vectors); identity_dofs: the subset whose rows are incomplete here (the outermost
 * node layer).  With overlap layers (ghost rows that are complete) the preconditioner is restricted additive Schwarz:
 * the residual is refreshed into the overlap before the local solve, the owners' part of the result is kept.
 * Call before the first Jacobian assembly. */
int fsi_set_partition(FsiCtx* ctx, int64_t num_owned_cells, int64_t n_ghost, const int64_t* ghost_dofs,
                      int64_t n_identity, const int64_t* identity_dofs, int64_t n_send, const int64_t* send_dofs,
                      double* sendbuf_dev, double* recvbuf_dev, const FsiComm* comm);

/* Collectives issued by the library itself (the default of vasp_amd/partition.py on the nccl backend with more than one
 * rank; VASPFSI_RCCL=0 keeps the callbacks): after fsi_set_partition, hand the
 * library an RCCL communicator and it queues ncclAllReduce (Krylov coefficient vectors in device memory) and grouped
 * ncclSend / ncclRecv (halo) on its solver stream; the FsiComm callbacks are then no longer called.  Counterpart of the MPI
 * reductions inside the reference's solver run [REF docs/simulation.md:14-32; simulation_common.py:217-220].
 * fsi_rccl_unique_id: 128 bytes from ncclGetUniqueId, made by one rank and broadcast by the caller.  send_counts[p] /
 * recv_counts[p]: doubles exchanged with rank p, in the order of the partition's send / ghost lists (peers ascending).
 * RCCL is resolved with dlopen at the first call; FSI_ERR_DEVICE if it is not available.  id128 == NULL: destroy the
 * library's communicator and return to the FsiComm callbacks (what the caller does when any rank failed to set it up).
 * A RCCL call that fails later aborts the communicator (ncclCommAbort) and every further collective of the context returns
 * FSI_ERR_DEVICE at once, so that no rank waits for one that has left. */
int fsi_rccl_unique_id(void* id128);
int fsi_set_rccl(FsiCtx* ctx, const void* id128, int32_t rank, int32_t world, const int64_t* send_counts,
                 const int64_t* recv_counts);

/* ---- state / introspection (checkpoint, restart, parity dumps) --------------------------------------- */
/* which: 0 = dvp_["n"], 1 = dvp_["n-1"], 2 = last rhs b, 3 = last update du.  User layout, length ndof. */
int fsi_get_state(FsiCtx* ctx, int which, double* out);
int fsi_set_state(FsiCtx* ctx, int which, const double* in);
/* out[i] = state[dofs[i]] (user layout): what a hook needs of dvp_ on a patch (the inlet facets of
 * assemble(inner(v, n) * ds(inlet)) [REF src/vasp/simulations/simulation_common.py:276]) without the whole vector
 * crossing PCIe every time step. */
int fsi_get_values(FsiCtx* ctx, int which, int64_t n, const int64_t* dofs, double* out);
int64_t fsi_num_dofs(const FsiCtx* ctx);
int64_t fsi_matrix_nnz(const FsiCtx* ctx);
/* Free and total bytes of the context's device as the HIP runtime of the library sees them (hipMemGetInfo): how much of the
 * 288 GB one context of a given mesh takes, asked without a second runtime in the process. */
int fsi_device_memory(FsiCtx* ctx, int64_t* free_bytes, int64_t* total_bytes);
/* Copies the assembled matrix out in user-layout row/column numbering (CSR, rows sorted).  rowptr [ndof+1],
 * cols/vals [nnz].  Values are the un-equilibrated Jacobian after ident_zeros and bc.apply. */
int fsi_get_matrix(FsiCtx* ctx, int64_t* rowptr, int64_t* cols, double* vals);
/* z = M^-1 r: one application of the active preconditioner (precond 0: the field-split block preconditioner with its fixed
 * sweep counts - a fixed linear operator; 1: ILU(0)) to a residual in the user layout, as fsi_solve's Krylov method applies
 * it - the PCApply of a PETSc binding, and what the tests check linearity and the quality of the approximation on.  Single
 * contexts only. */
int fsi_apply_preconditioner(FsiCtx* ctx, const double* r, double* z);
/* y = A x with the device SpMV (user layout in/out). */
int fsi_spmv(FsiCtx* ctx, const double* x, double* y);

/* ---- per-step diagnostics of post_solve on the device --------------------------------------------------- */
/* Replaces: peval / print_probe_points / print_solid_probe_points [REF src/vasp/simulations/simulation_common.py:157-222].
 * Points are located once on the host (cell + barycentric coordinates, -1 = outside: the reference's +inf sentinel is
 * the caller's business); out[i] = d_x d_y d_z v_x v_y v_z p of dvp_["n"] at point i (P2 / P1 interpolation). */
int fsi_probe(FsiCtx* ctx, int64_t n, const int32_t* cells, const double* bary, double* out);
/* Replaces: the DG0 projections of calculate_and_print_flow_properties (:253-317) and compute_minimum_jacobian (:320-348):
 * out[0..3] = mean, min, max over the cells of the cell-mean |v|, and min over the cells of the cell-mean det(I + grad d).
 * After fsi_set_partition: over the cells this context owns (the caller combines the ranks with the owned-cell counts). */
int fsi_flow_stats(FsiCtx* ctx, double* out);

/* ---- post-processing kernels on the resident state ------------------------------------- */
/* Replaces the element loop of compute_stress_strain [REF src/vasp/postprocessing/postprocessing_fenics/
 * compute_stress_strain.py:188-263]: for each listed SOLID cell (index in the mesh handed to fsi_create) the DG1
 * coefficients (one per local vertex) of the L2-projected Cauchy stress 1/J F S F^T and Green-Lagrange strain E of
 * dvp_["n"]'s displacement, and of the projected largest principal value of each.
 * out[i][80] = TrueStress[4][3][3], GreenLagrangeStrain[4][3][3], MaxPrincipalStress[4], MaxPrincipalStrain[4]. */
int fsi_stress_strain(FsiCtx* ctx, int64_t n, const int32_t* cells, double* out);
/* Replaces Stress.__call__ of compute_hemodynamics [REF .../compute_hemodynamics.py:91-157]: tangential traction
 * Ft = F - (F.n) n, F = -2 mu sym(grad v) n, on the listed exterior facets (cell + local index of the opposite vertex),
 * projected with the surface mass matrix onto the DG1 space of each boundary cell; out[f][3][3] = value at the three
 * facet vertices (local vertices of the cell in ascending order without the opposite one) x component. */
int fsi_wall_shear_stress(FsiCtx* ctx, int64_t nf, const int32_t* facet_cells, const int32_t* facet_local, double mu,
                          double* out);

/* ---- timing of the device kernels (HIP events on the solver stream) ---------------------------------- */
typedef struct FsiTimers {
  double residual_ms;  int64_t residual_calls;
  double jacobian_ms;  int64_t jacobian_calls;
  double factor_ms;    int64_t factor_calls;
  double spmv_ms;      int64_t spmv_calls;
  double precond_ms;   int64_t precond_calls;
  double ortho_ms;     int64_t ortho_calls;
  double krylov_ms;    int64_t krylov_solves;   int64_t krylov_iters;
  int64_t inner_vv_iters;                    /* inner BiCGStab iterations of the block preconditioner: velocity block */
  int64_t inner_schur_iters;                 /* ... pressure Schur complement                                          */
  int64_t inner_dd_iters;                    /* ... displacement block                                                 */
  int64_t precond_applies;
  double solid_spmv_ms;  int64_t solid_spmv_calls;   /* sampled launches of the solid-block SpMV (Chebyshev sweeps)   */
  int64_t solid_nnz;     int64_t solid_rows;         /* size of that block                                            */
  double db_spmv_ms;     int64_t db_spmv_calls;      /* sampled launches of the FP32 component-diagonal SpMV          */
  int64_t db_pairs;      int64_t db_nodes;           /* node pairs / nodes of that structure                          */
  double sc_spmv_ms;     int64_t sc_spmv_calls;      /* sampled launches of the scalar-ratio displacement SpMV        */
  int64_t disp_scalar;                               /* bit 0: displacement sweeps use that kernel; bit 1: LDS-tiled  */
  int64_t tile_entries;                              /* total length of the per-tile distinct-neighbour lists         */
  /* orthogonalisation of the recycled GCR: columns of Q / of the direction store streamed since the last reset, and the
     number of launches that streamed them (bytes = columns x ld x element size; see csrc/fsi_gcr.hip)                 */
  int64_t ortho_q_cols;  int64_t ortho_q_launches;
  int64_t ortho_z_cols;  int64_t ortho_z_launches;
  int64_t q_elem_bytes;                              /* 4: Q stored in FP32, 8: FP64                                   */
  int64_t ldq;           int64_t ldz;                /* column strides of Q and of the direction store (elements)     */
  int64_t krylov_dirs;   int64_t krylov_cap;         /* directions currently kept / capacity                          */
  int64_t schur_nnz;     int64_t schur_rows;         /* explicit Schur complement                                     */
  double flush_ms;       int64_t flush_calls;        /* k_gcr_flush (one pass over the direction store per solve)     */
  double schur_ms;       int64_t schur_calls;        /* sampled launches of the Schur-complement sweep                */
  int64_t schur_elem_bytes;                          /* value bytes next to a 4-byte column: 8 FP64, 4 FP32; 0: FP16
                                                        value and 16-bit tile-local column packed in the 4 bytes     */
  int64_t node_pairs;    int64_t node_vertex_pairs;  /* P2 node pairs / node-vertex pairs of the matrix graph         */
  int64_t spmv_fp32_calls;                           /* outer products that ran on the FP32 copy of the matrix        */
  int64_t sweep_flags;                               /* what the preconditioner sweeps actually run as (the context's
                                                        state, not the environment's): bit 0 tiled sweeps fused with
                                                        the Chebyshev update, 1 FP16 packed records, 2 solid block in
                                                        FP32, 3 solid sweeps fused (block Jacobi), 4 solid two-level
                                                        cycle ready, 5 displacement two-level cycle ready, 6 the
                                                        outer products take the displacement rows from their pair
                                                        form (FsiTuning.compact_drows, checked at the last refresh)   */
  int64_t part_allreduces;                           /* partitioned runs: all-reduces issued inside Krylov iterations
                                                        (FP64 basis: one per Gram-Schmidt pass; FP32 basis: two per pass
                                                        + one for its FP64 window; + one when an iteration looks
                                                        converged)                                                    */
  int64_t assembly_colours;                          /* colours of the assembly colouring (one launch of the residual /
                                                        Jacobian kernel each: bitwise reproducible scatter-adds); 0: one
                                                        launch over all cells with unordered atomics (FSI_ASSEMBLY=atomic) */
  /* what the linear solver had to do beyond iterating                                      */
  int64_t gcr_arnoldi_steps;                         /* directions made from the last q because the residual had not moved   */
  int64_t gcr_restarts;                              /* solves that dropped the kept directions (pairs lost / full store stalled) */
  int64_t newton_retries;                            /* Newton iterations whose solve failed on a stale Jacobian and succeeded
                                                        after a refresh                                                      */
  int64_t fp32_fallbacks;                            /* Jacobian lifetimes that lost the FP32 basis and finished in FP64      */
  int64_t verdicts_skipped;                          /* FP32 basis: loose answers (>= 1e-3) returned on the recurrence residual */
  int64_t reorth_forced;                             /* second Gram-Schmidt passes made because the first one showed the kept
                                                        columns non-orthonormal                                              */
  int64_t dd_cache_hits;                             /* Jacobian refreshes that found the displacement block unchanged (three
                                                        checksums) and kept its coarse operator and eigenvalue estimate      */
  int64_t newton_late_solves;                        /* Newton iterations solved with the late (tighter) forcing term        */
} FsiTimers;
/* ---- the solid cycle's exact coarse solve (round 5, csrc/fsi_bcr.hip): test and planning hooks ---------------------------- */
/* Host-only dry run of the planner on any symmetric vertex graph (no device): stats_out[8] = usable, blocks (BFS levels), largest
 * block [unknowns], reduction levels, operator bytes (FP32), set-up arena bytes (FP64), set-up flops, launches per solve; pos_out /
 * level_out (may be NULL): [nc] position in BFS-level order / BFS level of every node. */
int fsi_bcr_plan_graph(int64_t nc, const int64_t* cptr, const int32_t* ccol, int64_t* stats_out, int32_t* pos_out, int32_t* level_out);
/* out[12]: coarse nodes, 3x3 blocks, planned, ready (operators of the current Jacobian), BFS blocks, reduction levels, operator
 * bytes, launches per solve, largest block, solves so far, set-up flops per refresh, two-level cycle ready */
int fsi_solid_coarse_info(const FsiCtx* ctx, int64_t* out);
/* the coarse level's block-CSR operator as the sweeps / the reduction see it: cptr[nc + 1], ccol[nblk], cvals[9 nblk] */
int fsi_solid_coarse_matrix(FsiCtx* ctx, int64_t* cptr, int32_t* ccol, float* cvals);
/* x = A_c^-1 rhs by the production kernels (3 doubles per coarse node) */
int fsi_solid_coarse_solve(FsiCtx* ctx, const double* rhs, double* x);

/* Host-only test hook of the XCD-aware workgroup order (csrc/fsi_kernels.hpp: xcd_unit / xcd_span) that k_residual and the sweep
 * kernels use: unit_out[span] = the unit (tile, cell pair, row block) logical workgroup L of a launch over n units works on, -1 for
 * a workgroup without one; returns span = the number of logical workgroups launched (unit_out may be NULL), -1 for n < 0. */
int64_t fsi_xcd_order(int64_t n, int64_t* unit_out);

int fsi_get_timers(FsiCtx* ctx, FsiTimers* out, int reset);
/* Run totals since fsi_create, without resolving the phase timers (no device synchronisation, safe to call every time step,
 * unaffected by fsi_get_timers(reset = 1)): out[0] newton_retries, out[1] fp32_fallbacks, out[2] gcr_restarts - what the product
 * driver prints as "Linear solver events so far" (the FsiTimers fields of the same names count since the last reset) -, out[3]
 * Newton solves whose tolerance came from the adaptive forcing term, out[4] times such a solve was tightened because its UNSCALED
 * residual was above the tolerance, out[5] exact coarse solves of the solid cycle, out[6], out[7] reserved (0). */
int fsi_get_solver_events(const FsiCtx* ctx, int64_t out[8]);
/* Measurement aid: streams `bytes` of the (idle) Krylov store once per kernel with 4-, 8-, 16- and 32-byte loads and
 * 4-, 8-, 16-byte stores per lane (kernels k_cal_read<...> / k_cal_write<...>), so that a rocprofv3 --pmc FETCH_SIZE /
 * WRITE_SIZE pass can be calibrated against known byte counts at the access widths the solver kernels use.  Discards
 * the recycled Krylov directions. */
int fsi_calibration_streams(FsiCtx* ctx, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VASPFSI_H */
