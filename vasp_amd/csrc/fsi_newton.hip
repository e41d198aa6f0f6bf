// The hot path behind the C-ABI: assemble(-F), assemble(J) + preconditioner refresh, one linear solve, and turtleFSI's
// quasi-Newton loop `newtonsolver` as VaSP drives it (solver keys REF src/vasp/simulations/offset_stenosis.py:44-48).
#include "fsi_host.hpp"

using namespace fsi;
using namespace fsi::host;

extern "C" {

int fsi_solver_setup(FsiCtx* ctx) {
  if (!ctx) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  Phase ph(ctx, &ctx->t_jac);
  HIPCHK(hipMemsetAsync(ctx->A_pre.p, 0, ctx->nnz * sizeof(double), ctx->stream));
  launch_jacobian(ctx->stream, PART_LINEAR, ctx->C, elem_arrays(ctx), elem_params(ctx), ctx->U.p, ctx->U1.p,
                  ctx->rowptr.p, ctx->nadj_ptr.p, ctx->A_pre.p, cell_colours(ctx));
  launch_add_at(ctx->stream, ctx->A_pre.p, ctx->rb_pos.p, ctx->rb_val.p, ctx->scheme.th0, ctx->nrobin);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->have_jacobian = false;
  return FSI_OK;
}

int fsi_assemble_residual(FsiCtx* ctx, double* norm) {
  if (!ctx) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  {
    Phase ph(ctx, &ctx->t_res);
    if (!ctx->Re.p) HIPCHK(hipMemsetAsync(ctx->F.p, 0, ctx->ndof * sizeof(double), ctx->stream));    // the gather writes every entry
    launch_residual(ctx->stream, ctx->C, elem_arrays(ctx), elem_params(ctx), ctx->U.p, ctx->U1.p, ctx->F.p, residual_gather(ctx));
    launch_add_indexed(ctx->stream, ctx->F.p, ctx->pf_dofs.p, ctx->pf_coef.p, ctx->P, ctx->npf);
    launch_robin_residual(ctx->stream, ctx->nrobin_rows, ctx->rb_urow.p, ctx->rb_ptr.p, ctx->rb_col.p, ctx->rb_val.p, ctx->scheme.th0,
                          ctx->scheme.th1, ctx->U.p, ctx->U1.p, ctx->F.p);
    launch_negate(ctx->stream, ctx->b.p, ctx->F.p, ctx->ndof);
    launch_bc_rhs(ctx->stream, ctx->b.p, ctx->U.p, ctx->bc_dofs.p, ctx->bc_vals.p, ctx->nbc);
    zero_ghost(ctx, ctx->b.p);
    HIPCHK(hipGetLastError());
  }
  double nrm = 0.0;
  FSICHK(gnorm2(ctx, ctx->b.p, &nrm));
  if (norm) *norm = nrm;
  return FSI_OK;
}

int fsi_assemble_jacobian(FsiCtx* ctx) {
  if (!ctx) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  {
    Phase ph(ctx, &ctx->t_jac);
    HIPCHK(hipMemsetAsync(ctx->A.p, 0, ctx->nnz * sizeof(double), ctx->stream));
    launch_jacobian(ctx->stream, PART_NONLINEAR, ctx->C, elem_arrays(ctx), elem_params(ctx), ctx->U.p, ctx->U1.p,
                    ctx->rowptr.p, ctx->nadj_ptr.p, ctx->A.p, cell_colours(ctx), ctx->tune.jacobian_waves, ctx->tune.jacobian_mfma);
    if (getenv("FSI_DEBUG")) { HIPCHK(hipStreamSynchronize(ctx->stream)); fprintf(stderr, "[fsi] jacobian kernel done\n"); fflush(stderr); }
    launch_matrix_finish(ctx->stream, ctx->ndof, ctx->rowptr.p, ctx->diagpos.p, ctx->A.p, ctx->A_pre.p, ctx->mbc_dofs.p,
                         ctx->nmbc, ctx->rowscale.p, ctx->iflags.p + 16);
    HIPCHK(hipGetLastError());
    if (getenv("FSI_DEBUG")) { HIPCHK(hipStreamSynchronize(ctx->stream)); fprintf(stderr, "[fsi] matrix finish done\n"); fflush(stderr); }
  }
  ctx->op32_ok = false;
  const bool copy32 = ctx->op32_policy && ctx->kry_fp32_policy != 0 && ctx->precond == 0 && ctx->A32.p;
  // the d rows in pair form for the outer products (FsiTuning.compact_drows): extracted from THIS matrix, and adopted only if the
  // kernel found nothing else in those rows (one flag read per refresh)
  ctx->drows_ok = false;
  if (getenv("FSI_DEBUG_DROWS_INJECT") && ctx->nnz > 1) {
    // test hook: a value where the forms leave a structural zero - entry 1 of the first d row (column d_y of node 0's first
    // neighbour in a d_x row) - so that the check below has a matrix to refuse
    const int64_t pos = 1;
    const double val = 0.125;
    int64_t* dpos = reinterpret_cast<int64_t*>(ctx->scratch.p + 4098);
    HIPCHK(hipMemcpyAsync(dpos, &pos, sizeof pos, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->scratch.p + 4099, &val, sizeof val, hipMemcpyHostToDevice, ctx->stream));
    launch_add_at(ctx->stream, ctx->A.p, dpos, ctx->scratch.p + 4099, 1.0, 1);
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (ctx->tune.compact_drows && ctx->N2 > 0) {
    const size_t npairs6 = 6 * (size_t)ctx->nadj.n;
    if (!ctx->Ad64.p) HIPCHK(ctx->Ad64.alloc(npairs6));
    if (copy32 && !ctx->Ad32.p) HIPCHK(ctx->Ad32.alloc(npairs6));
    HIPCHK(hipMemsetAsync(ctx->iflags.p + 20, 0, sizeof(int32_t), ctx->stream));
    launch_drows_extract(ctx->stream, ctx->N2, ctx->rowptr.p, ctx->A.p, ctx->nadj_ptr.p, ctx->Ad64.p, copy32 ? ctx->Ad32.p : nullptr, ctx->iflags.p + 20);
    int32_t found = 1;
    HIPCHK(hipMemcpyAsync(&found, ctx->iflags.p + 20, sizeof found, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->drows_ok = found == 0;
    if (getenv("FSI_DEBUG")) fprintf(stderr, "[fsi] displacement rows of the outer product in pair form: %s\n", ctx->drows_ok ? "yes" : "no (other entries found)");
  }
  if (copy32) {
    launch_pad_vals32(ctx->stream, ctx->N2, ctx->V, ctx->rowptr.p, ctx->A.p, ctx->a32_ptr.p, ctx->a32_ptail, ctx->a32_tail_nnz,
                      ctx->a32_tail_src, ctx->A32.p, ctx->drows_ok);       // rows are equilibrated: |entries| <= 1
    ctx->op32_ok = true;
  }
  gcr_reset(ctx);          // the recycled directions belong to the previous matrix
  for (double& h : ctx->nw_hist) h = 0.0;      // ... and so does what was learnt about the Newton iteration's contraction
  ctx->utol_ratio = 0.0;                       // ... and about the row scaling's effect on its residuals
  // what decides the storage precision of the basis belongs to the Jacobian that has just been replaced: the largest
  // right-hand side seen (one large early |b|, e.g. the first step from rest, must not keep tol_hint low for the whole run)
  // and a fall-back to FP64 after a failed cycle (a system that lost FP32 once may not lose it with the next matrix; after
  // two such failures the context stays FP64)
  ctx->bnorm_max = 0.0;
  if (ctx->kry_fp32_policy == 3 && ctx->kry_fp32_failures < 2) ctx->kry_fp32_policy = 2;
  ctx->have_jacobian = true;
  ctx->have_monolithic_lu = false;
  const int rc = refresh_preconditioner(ctx);
  if (getenv("FSI_DEBUG_FORCE_PREC_BAD")) ctx->prec_bad = true;     // test hook: this rank's self-test "fails"
  if (!ctx->part) return rc;
  // the self-test and the pivot checks above are rank-local: all ranks leave with the same verdict, so that either all
  // of them enter the collectives of the next solve or none does
  const bool bad = ctx->precond == 0 && ctx->prec_bad;
  const int all = agree(ctx, (rc != FSI_OK || bad) ? FSI_ERR_LINEAR : FSI_OK);
  if (all == FSI_ERR_DEVICE) return all;
  if (all != FSI_OK) ctx->prec_bad = true;      // fsi_solve reports it on every rank
  return rc;
}

int fsi_solve(FsiCtx* ctx, double lin_rtol, int32_t lin_max_it, int32_t lin_solver, int32_t* iters, double* relres) {
  if (!ctx) return FSI_ERR_INVALID;
  if (!ctx->have_jacobian) { ctx->err = "fsi_solve: no Jacobian assembled"; return FSI_ERR_INVALID; }
  if (ctx->precond == 0 && ctx->prec_bad) { ctx->err = "block preconditioner: Chebyshev sweeps diverge on this Jacobian"; return FSI_ERR_LINEAR; }
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->part && lin_solver == 1) { ctx->err = "fsi_solve: the partitioned path runs GCR only (lin_solver 0)"; return FSI_ERR_INVALID; }
  launch_mul(ctx->stream, ctx->bs.p, ctx->rowscale.p, ctx->b.p, ctx->ndof);
  int it = 0;
  double rr = 0.0;
  int rc;
  {
    Phase ph(ctx, &ctx->t_kry);
    if (lin_solver == 1) {
      launch_copy(ctx->stream, ctx->F.p, ctx->bs.p, ctx->ndof);     // F is free between residual assemblies
      rc = solve_bicgstab(ctx, ctx->F.p, ctx->du.p, lin_rtol, lin_max_it, &it, &rr);
    } else {
      rc = solve_gcr(ctx, ctx->bs.p, ctx->du.p, lin_rtol, lin_max_it, &it, &rr);
    }
  }
  if (iters) *iters = it;
  if (relres) *relres = rr;
  return rc;
}

int fsi_newton_solve(FsiCtx* ctx, const FsiNewtonOpts* o, FsiNewtonIter* iters, int32_t* n_iters) {
  if (!ctx || !o || !iters || !n_iters) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  int it = 0;
  double residual = 1e8, rel_res = 1e8, last_residual = 1e8;
  double prev_eta = 0.0;           // tolerance of the previous iteration's solve (adaptive forcing term below)
  *n_iters = 0;
  while (rel_res > o->rtol && residual > o->atol && it < o->max_it) {
    const bool rec = (it == 0 && o->recompute_tstep > 0 && o->counter % o->recompute_tstep == 0) ||
                     (it > 0 && o->recompute > 0 && it % o->recompute == 0) || (it > 0 && last_residual < residual) ||
                     (it == 0 && o->counter == o->first_step_num) || !ctx->have_jacobian;
    if (rec) FSICHK(fsi_assemble_jacobian(ctx));
    double bnorm = 0.0;
    FSICHK(fsi_assemble_residual(ctx, &bnorm));
    last_residual = residual;
    int32_t lit = 0;
    double lrr = 0.0;
    // inexact Newton: the update only has to push the residual three orders below the Newton tolerance, never tighter
    // than lin_rtol; without this the last iteration of every step solves a 1e-10-sized system to 1e-20
    double eta = o->lin_rtol;
    if (bnorm > 0.0 && o->atol > 0.0) eta = std::max(eta, std::min(1e-2, ctx->newton_forcing * o->atol / bnorm));
    // What a quasi-Newton iteration can gain is bounded by its (stale) Jacobian, not by its linear solve: on the bench the first
    // iteration of a time step was solved to 2e-6 and contracted the residual by 3e-3, step after step (and the aneurysm file at
    // its own tolerances: solved to 1e-9, contracted by 1e-3).  The contraction the iteration of the SAME INDEX reached one time
    // step ago, under the same Jacobian, is known; a linear residual of 0.3 x that (`newton_adaptive`) cannot show in the next
    // nonlinear residual.  What is remembered is the contraction net of the linear tolerance that was allowed (rho_obs - eta: the
    // two add at worst), so a looser solve cannot feed back into a looser solve.  Never looser than 1e-2, never applied to an
    // iteration that refreshes the Jacobian (nothing is known about the new one), and the late rule below still tightens what is
    // likely the last solve of the step.
    if (it > 0 && it <= 8 && bnorm > 0.0 && residual > 0.0 && residual < 1e7) {
      const double rho_obs = bnorm / residual;                      // |b_it| / |b_{it-1}| of THIS step
      ctx->nw_hist[it - 1] = std::max(rho_obs - prev_eta, 0.5 * rho_obs);
    }
    // (newton_forcing = 0 means every system to lin_rtol, the direct-LU policy: nothing is loosened then)
    if (ctx->newton_adaptive > 0.0 && ctx->newton_forcing > 0.0 && !rec && it < 8 && ctx->nw_hist[it] > 0.0 && ctx->nw_hist[it] < 1.0) {
      const double eta_a = std::min(1e-2, ctx->newton_adaptive * ctx->nw_hist[it]);
      if (eta_a > eta) {
        // the contraction was observed on the UNSCALED |b| (the norm the policy's decisions hang on); the Krylov method stops on
        // the row-equilibrated norm.  solve_gcr holds the answer to both: eta_a in its own norm, and an unscaled residual below
        // eta_a |b| - tightening towards the non-adaptive tolerance while that fails (known-answer case, dt = 0.01: a scaled 1e-3
        // left 0.4 |b| in the penalty rows, the residual rose and the policy refreshed a Jacobian the reference keeps)
        ctx->utol = eta_a; ctx->utol_rtol_floor = eta; ctx->b_unscaled = bnorm;
        eta = eta_a;
        ctx->newton_adaptive_solves += 1;
      }
    }
    // Late iterations - the previous update was already within `late_factor` of the stopping tolerance, so this one is
    // likely the last of the step - are solved with the tighter forcing term: what an inexact LAST solve leaves in the state is
    // what separates the run from the reference's direct-LU trajectory (production defaults against
    // exact solves).  Only while |b| is so far below the largest right-hand side of this Jacobian's life that the tighter
    // tolerance stays above the floor the storage precision of the Krylov basis was chosen for (tol_hint below).
    const double f_late = ctx->newton_forcing_late;
    if (it > 0 && f_late > 0.0 && f_late < ctx->newton_forcing && bnorm > 0.0 && o->atol > 0.0 &&
        (rel_res <= ctx->newton_late_factor * o->rtol || bnorm <= ctx->newton_late_factor * o->atol) &&
        bnorm <= (f_late / ctx->newton_forcing) * ctx->bnorm_max) {
      eta = std::max(o->lin_rtol, std::min(eta, f_late * o->atol / bnorm));
      ctx->newton_late_solves += 1;
    }
    // the tightest linear tolerance this Newton policy can ask for while the present Jacobian lives: its forcing term at
    // the largest right-hand side seen so far (decides the storage precision of the Krylov basis, see solve_gcr)
    ctx->bnorm_max = std::max(ctx->bnorm_max, bnorm);
    ctx->tol_hint = o->lin_rtol;
    if (ctx->bnorm_max > 0.0 && o->atol > 0.0 && ctx->newton_forcing > 0.0)
      ctx->tol_hint = std::max(o->lin_rtol, std::min(1e-2, ctx->newton_forcing * o->atol / ctx->bnorm_max));
    // in_newton / tol_hint hold for the two solves below and for nothing else: the guard clears them on EVERY way out of this
    // iteration (the early return of the refresh-and-retry path included), so that a later direct fsi_solve on this context is
    // never taken for a solve inside Newton - it would skip the FP64 verdict and choose the basis precision from a stale hint
    struct NewtonScope { FsiCtx* c; ~NewtonScope() { c->tol_hint = 0.0; c->in_newton = false; c->utol = 0.0; } } newton_scope{ctx};
    ctx->in_newton = true;
    prev_eta = eta;
    int src = fsi_solve(ctx, eta, o->lin_max_it, o->lin_solver, &lit, &lrr);
    bool rec_retry = false;
    if (src == FSI_ERR_LINEAR && !rec && !ctx->prec_bad) {
      // The iteration did not converge with a Jacobian (and a preconditioner, and a recycled space) that other states made:
      // what the reference's policy does when the residual grows - assemble the Jacobian at the present state - is done
      // here for the linear solver's sake, once, and the system is solved again; the iteration is reported as a refresh.
      FSICHK(fsi_assemble_jacobian(ctx));
      int32_t lit2 = 0;
      src = fsi_solve(ctx, eta, o->lin_max_it, o->lin_solver, &lit2, &lrr);
      lit += lit2;
      rec_retry = true;
      ctx->newton_retries += 1;
    }
    ctx->tol_hint = 0.0;
    ctx->in_newton = false;
    FSICHK(src);
    launch_axpy(ctx->stream, ctx->U.p, o->lmbda, ctx->du.p, ctx->ndof);
    launch_bc_set(ctx->stream, ctx->U.p, ctx->bc_dofs.p, ctx->bc_vals.p, ctx->nbc);
    residual = bnorm;
    // "r (rel)": L2(Omega) function norm of the update, as dolfin.norm(Function, 'l2') in the reference's newtonsolver
    HIPCHK(hipMemsetAsync(ctx->scratch.p + 4097, 0, sizeof(double), ctx->stream));
    if ((ctx->part ? ctx->C_owned : ctx->C) > 0)
      launch_l2norm(ctx->stream, ctx->part ? ctx->C_owned : ctx->C, elem_arrays(ctx), ctx->du.p, ctx->scratch.p, ctx->scratch.p + 4097);
    FSICHK(host_scalar(ctx, ctx->scratch.p + 4097, &rel_res));
    FSICHK(allreduce(ctx, &rel_res, 1));
    rel_res = std::sqrt(rel_res);
    iters[it] = FsiNewtonIter{residual, rel_res, (rec || rec_retry) ? 1 : 0, lit, lrr};
    if (getenv("FSI_DEBUG")) fprintf(stderr, "[fsi] newton %d: |b| %.3e |du|_L2 %.3e refresh %d retry %d krylov %d relres %.2e eta %.1e\n", it, residual, rel_res, (int)rec, (int)rec_retry, (int)lit, lrr, eta);
    it += 1;
    *n_iters = it;
    if (!(residual <= 1e20) || !(rel_res <= 1e20)) {
      ctx->err = "Error: The simulation has diverged during the Newton solve.";
      return FSI_ERR_DIVERGED;
    }
  }
  return FSI_OK;
}

int fsi_shift(FsiCtx* ctx) {
  if (!ctx) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  launch_copy(ctx->stream, ctx->U1.p, ctx->U.p, ctx->ndof);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return FSI_OK;
}

}  // extern "C"
