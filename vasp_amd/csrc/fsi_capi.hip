// C-ABI of libvaspfsi.so (include/vaspfsi.h): problem set-up, Newton driver, Krylov methods.
//
// Host-side logic follows turtleFSI's monolithic.py / newtonsolver.py as VaSP uses them; the
// arithmetic runs in the HIP kernels of fsi_assembly.hip / fsi_solver.hip.
#include "fsi_host.hpp"

using namespace fsi;
using namespace fsi::host;

namespace {

const int TET_EDGES[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};

}  // namespace

namespace fsi {
namespace host {

ElemArrays elem_arrays(FsiCtx* c) {
  return ElemArrays{c->geom.p, c->cell_dofs.p, c->cell_kind.p, c->cell_region.p, c->cell_rank.p, c->cell_prow.p, c->enbr.p, c->epnbr.p};
}
ResidualGather residual_gather(const FsiCtx* c) {
  ResidualGather rg;
  if (c->Re.p) { rg.Re = c->Re.p; rg.N2 = c->N2; rg.V = c->V; rg.inc_ptr = c->inc_ptr.p; rg.inc = c->inc.p; rg.pinc_ptr = c->pinc_ptr.p; rg.pinc = c->pinc.p; }
  return rg;
}
CellColours cell_colours(const FsiCtx* c) {
  CellColours cc;
  if (c->ncellcol > 0) { cc.ncolours = c->ncellcol; cc.cells = c->col_cells.p; cc.ptr = c->h_col_ptr.data(); }
  return cc;
}
ElemParams elem_params(FsiCtx* c) {
  ElemParams ep;
  ep.sc = c->scheme;
  for (int i = 0; i < MAX_REGIONS; ++i) { ep.fluid[i] = c->fluid[i]; ep.solid[i] = c->solid[i]; }
  return ep;
}

int host_scalar(FsiCtx* ctx, const double* dptr, double* out) {
  HIPCHK(hipMemcpyAsync(out, dptr, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return FSI_OK;
}
int dot_n(FsiCtx* ctx, const double* x, const double* y, int64_t n, double* out) {
  launch_dot(ctx->stream, x, y, n, ctx->scratch.p, ctx->scratch.p + 4096);
  return host_scalar(ctx, ctx->scratch.p + 4096, out);
}
int dot(FsiCtx* ctx, const double* x, const double* y, double* out) { return dot_n(ctx, x, y, ctx->ndof, out); }
int norm2(FsiCtx* ctx, const double* x, double* out) {
  FSICHK(dot(ctx, x, x, out));
  *out = std::sqrt(*out);
  return FSI_OK;
}

// ---- element partition: sums over ranks, owner -> ghost refresh (fsi_set_partition) -----------------------------------
int allreduce(FsiCtx* ctx, double* v, int n) {
  if (!ctx->part) return FSI_OK;
  ctx->allreduce_calls += 1;
  if (ctx->rccl) return rccl_allreduce_host(ctx, v, n);
  if (ctx->comm.allreduce_sum(ctx->comm.user, v, n) != 0) { ctx->err = "allreduce callback failed"; return FSI_ERR_DEVICE; }
  return FSI_OK;
}
// Partitioned runs: a status decided from rank-local data (a preconditioner self-test, a pivot, a device error) must be
// the same on every rank before the next collective, or the job hangs in it.  Every rank calls this at the same points.
int agree(FsiCtx* ctx, int rc) {
  if (!ctx->part) return rc;
  double bad = rc == FSI_OK ? 0.0 : 1.0;
  if (ctx->rccl) { if (rccl_allreduce_host(ctx, &bad, 1) != FSI_OK) return FSI_ERR_DEVICE; }
  else if (ctx->comm.allreduce_sum(ctx->comm.user, &bad, 1) != 0) { ctx->err = "allreduce callback failed"; return FSI_ERR_DEVICE; }
  ctx->allreduce_calls += 1;
  if (rc == FSI_OK && bad > 0.0) { ctx->err = "another rank of the partitioned job reported an error"; return FSI_ERR_LINEAR; }
  return rc;
}
// dot / norm over all ranks; the operands carry zeros in their ghost entries, so the local sums add up
int gdot(FsiCtx* ctx, const double* x, const double* y, double* out) {
  FSICHK(dot(ctx, x, y, out));
  return allreduce(ctx, out, 1);
}
int gnorm2(FsiCtx* ctx, const double* x, double* out) {
  FSICHK(gdot(ctx, x, x, out));
  *out = std::sqrt(*out);
  return FSI_OK;
}
int halo_update(FsiCtx* ctx, double* x) {
  if (!ctx->part) return FSI_OK;
  ctx->halo_calls += 1;
  if (ctx->nsend) launch_gather(ctx->stream, ctx->sendbuf, x, ctx->send_idx.p, ctx->nsend);
  if (ctx->rccl) {      // pack -> grouped send / recv -> unpack, all on the solver stream: the host does not wait
    FSICHK(rccl_halo(ctx));
    if (ctx->nghost) launch_scatter(ctx->stream, x, ctx->recvbuf, ctx->ghost_idx.p, ctx->nghost);
    return FSI_OK;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->comm.halo_exchange(ctx->comm.user) != 0) { ctx->err = "halo_exchange callback failed"; return FSI_ERR_DEVICE; }
  if (ctx->nghost) launch_scatter(ctx->stream, x, ctx->recvbuf, ctx->ghost_idx.p, ctx->nghost);
  return FSI_OK;
}
void zero_ghost(FsiCtx* ctx, double* x) {
  if (ctx->part && ctx->nghost) launch_bc_set(ctx->stream, x, ctx->ghost_idx.p, ctx->ghost_zero.p, ctx->nghost);
}
int rebuild_matrix_bc(FsiCtx* ctx) {
  std::vector<int32_t> m(ctx->h_bc);
  m.insert(m.end(), ctx->h_ident.begin(), ctx->h_ident.end());
  ctx->nmbc = (int64_t)m.size();
  return upload(ctx, ctx->mbc_dofs, m);
}

}  // namespace host
}  // namespace fsi

// =========================================================================================================
extern "C" {

int fsi_set_chebyshev(FsiCtx* ctx, int32_t its_solid, double kappa_solid, int32_t its_fluid, double kappa_fluid,
                      int32_t its_schur, double kappa_schur, int32_t its_disp, double kappa_disp) {
  if (!ctx) return FSI_ERR_INVALID;
  if (its_disp > 0) ctx->cheb_its_d = its_disp;
  if (kappa_disp > 1.0) ctx->cheb_kappa_d = kappa_disp;
  if (its_solid > 0) ctx->cheb_its_s = its_solid;
  if (kappa_solid > 1.0) ctx->cheb_kappa_s = kappa_solid;
  if (its_fluid > 0) ctx->cheb_its_f = its_fluid;
  if (kappa_fluid > 1.0) ctx->cheb_kappa_f = kappa_fluid;
  if (its_schur > 0) ctx->cheb_its_p = its_schur;
  if (kappa_schur > 1.0) ctx->cheb_kappa_p = kappa_schur;
  gcr_reset(ctx);     // the recycled directions were built with another (fixed) preconditioner
  return FSI_OK;
}

int fsi_set_newton_forcing(FsiCtx* ctx, double forcing) {
  if (!ctx || !(forcing >= 0.0)) return FSI_ERR_INVALID;
  ctx->newton_forcing = forcing;
  return FSI_OK;
}

int fsi_set_linear_solver(FsiCtx* ctx, int32_t precond) {
  if (!ctx || precond < 0 || precond > 1) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  const bool changed = precond != ctx->precond;
  ctx->precond = precond;
  if (changed && ctx->have_jacobian) { gcr_reset(ctx); return refresh_preconditioner(ctx); }
  return FSI_OK;
}


const char* fsi_last_error(const FsiCtx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int64_t fsi_num_dofs(const FsiCtx* ctx) { return ctx ? ctx->ndof : 0; }
int64_t fsi_matrix_nnz(const FsiCtx* ctx) { return ctx ? ctx->nnz : 0; }

int fsi_device_memory(FsiCtx* ctx, int64_t* free_bytes, int64_t* total_bytes) {
  if (!ctx || !free_bytes || !total_bytes) return FSI_ERR_INVALID;
  size_t f = 0, t = 0;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemGetInfo(&f, &t));
  *free_bytes = (int64_t)f;
  *total_bytes = (int64_t)t;
  return FSI_OK;
}

int fsi_destroy(FsiCtx* ctx) {
  if (!ctx) return FSI_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  DevBuf<double>* dbl[] = {&ctx->geom, &ctx->A_pre, &ctx->A, &ctx->LU, &ctx->rowscale, &ctx->U, &ctx->U1, &ctx->F, &ctx->b,
                           &ctx->du, &ctx->bs, &ctx->tmp1, &ctx->tmp2, &ctx->tmp3, &ctx->tmp4, &ctx->tmp5, &ctx->tmp6,
                           &ctx->tmp7, &ctx->scratch, &ctx->bc_vals, &ctx->pf_coef, &ctx->rb_val, &ctx->KZ, &ctx->hcoef,
                           &ctx->gcr_out, &ctx->gcr_y, &ctx->gcr_cn, &ctx->KQh, &ctx->hcoef_hot};
  for (auto* b : dbl) b->release();
  ctx->KQ.release(); ctx->A32.release(); ctx->a32_ptr.release(); ctx->a32_cols.release(); ctx->Ad64.release(); ctx->Ad32.release();
  ctx->gcr_slots.release();
  ctx->gv_idx.release();
  if (ctx->gcr_host) { (void)hipHostFree(ctx->gcr_host); ctx->gcr_host = nullptr; }
  DevBuf<int32_t>* i32[] = {&ctx->user2solver, &ctx->solver2user, &ctx->cell_dofs, &ctx->cell_kind, &ctx->cell_region,
                            &ctx->cell_rank, &ctx->cell_prow, &ctx->nadj, &ctx->padj, &ctx->cols, &ctx->iflags, &ctx->bc_dofs,
                            &ctx->pf_dofs, &ctx->rb_row, &ctx->rb_col, &ctx->rb_urow, &ctx->rb_ptr, &ctx->col_cells, &ctx->inc, &ctx->pinc};
  for (auto* b : i32) b->release();
  ctx->Re.release(); ctx->inc_ptr.release(); ctx->pinc_ptr.release();
  DevBuf<int64_t>* i64[] = {&ctx->nadj_ptr, &ctx->padj_ptr, &ctx->rowptr, &ctx->diagpos, &ctx->rb_pos};
  for (auto* b : i64) b->release();
  ctx->sbmg_par.release(); ctx->sbmg_ccol.release(); ctx->sbmg_child.release(); ctx->sbmg_cfine.release(); ctx->sbmg_pw.release();
  ctx->sbmg_chw.release(); ctx->sbmg_cvals.release(); ctx->sbmg_cbinv12.release(); ctx->sbmg_work.release(); ctx->sbmg_cptr.release();
  ctx->sbmg_chptr.release(); ctx->sbmg_flag.release(); ctx->sbmg_cflag.release();
  rccl_destroy(ctx);
  bcr_free(ctx);
  ctx->s_vals32.release(); ctx->s_rec.release(); ctx->s_dinv.release();
  ctx->s_ploc.release(); ctx->s_tile_uptr.release(); ctx->s_tile_ulist.release();
  ctx->fs_rows.release(); ctx->fs_col.release(); ctx->fs_ptr.release(); ctx->fs_src.release();
  ctx->mg_par.release(); ctx->mg_ccol.release(); ctx->mg_child.release(); ctx->mg_cfine.release(); ctx->mg_pw.release();
  ctx->mg_chw.release(); ctx->mg_cptr.release(); ctx->mg_chptr.release(); ctx->mg_Ac.release(); ctx->mg_cc.release();
  ctx->mg_d0.release(); ctx->mg_dcinv4.release(); ctx->mg_cones.release(); ctx->mg_work.release(); ctx->mg_cflag.release();
  ctx->ghost_idx.release(); ctx->ident_idx.release(); ctx->send_idx.release(); ctx->mbc_dofs.release(); ctx->ghost_zero.release();
  ctx->enbr.release();
  ctx->epnbr.release();
  ctx->cellvals.release();
  for (auto* b : {&ctx->Adv, &ctx->Avp, &ctx->Apv, &ctx->App, &ctx->blk, &ctx->Mdd.vals, &ctx->Mvv.vals, &ctx->mask_s, &ctx->mask_f, &ctx->ss_vals, &ctx->dd_db, &ctx->vv_db, &ctx->adv_db, &ctx->s_vals}) b->release();
  ctx->s_rowptr.release(); ctx->s_diagpos.release(); ctx->s_cols.release();
  for (auto* b : {&ctx->snode, &ctx->ss_cols, &ctx->sb_col, &ctx->sb_row, &ctx->sb_stride}) b->release();
  ctx->sb_ptr.release(); ctx->sb_src.release(); ctx->sb_vals.release(); ctx->sb_dinv.release();
  ctx->sb_binv12.release(); ctx->sb_binv9.release();
  ctx->dd_db32.release(); ctx->vv_db32.release(); ctx->dd_dinv32.release(); ctx->vvf_dinv32.release();
  ctx->adv_rowmask.release(); ctx->vv_dinv.release(); ctx->Avp32.release(); ctx->Apv32.release(); ctx->dd_chat.release(); ctx->ones32.release(); ctx->dd_rowflag.release(); ctx->dd_rec.release(); ctx->vv_rec.release(); ctx->sb_rec.release();
  ctx->tile_ploc.release(); ctx->tile_uptr.release(); ctx->tile_ulist.release();
  for (auto* b : {&ctx->ss_rowptr, &ctx->ss_diagpos, &ctx->ss_src}) b->release();
  for (auto* b : {&ctx->node_solid, &ctx->vrank, &ctx->cols3, &ctx->cols_vp, &ctx->cols_pv, &ctx->cols_pp}) b->release();
  for (auto* b : {&ctx->rowptr3, &ctx->diagpos3, &ctx->rowptr_vp, &ctx->rowptr_pv, &ctx->rowptr_pp, &ctx->diagpos_pp}) b->release();
  for (int k = 0; k < 8; ++k) { if (ctx->ss_ev0[k]) (void)hipEventDestroy(ctx->ss_ev0[k]); if (ctx->ss_ev1[k]) (void)hipEventDestroy(ctx->ss_ev1[k]); }
  for (int k = 0; k < 8; ++k) { if (ctx->db_ev0[k]) (void)hipEventDestroy(ctx->db_ev0[k]); if (ctx->db_ev1[k]) (void)hipEventDestroy(ctx->db_ev1[k]); }
  for (int k = 0; k < 4; ++k) { if (ctx->sc_ev0[k]) (void)hipEventDestroy(ctx->sc_ev0[k]); if (ctx->sc_ev1[k]) (void)hipEventDestroy(ctx->sc_ev1[k]); }
  for (int k = 0; k < 4; ++k) { if (ctx->sch_ev0[k]) (void)hipEventDestroy(ctx->sch_ev0[k]); if (ctx->sch_ev1[k]) (void)hipEventDestroy(ctx->sch_ev1[k]); }
  for (PhaseTimer* t : {&ctx->t_res, &ctx->t_jac, &ctx->t_fac, &ctx->t_spmv, &ctx->t_prec, &ctx->t_ortho, &ctx->t_flush, &ctx->t_kry, &ctx->t_ss, &ctx->t_db, &ctx->t_sc})
    for (int k = 0; k < PhaseTimer::RING; ++k) { if (t->e0[k]) (void)hipEventDestroy(t->e0[k]); if (t->e1[k]) (void)hipEventDestroy(t->e1[k]); }
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (hipEvent_t e : {ctx->ev_split, ctx->ev_solid, ctx->ev_b}) if (e) (void)hipEventDestroy(e);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return FSI_OK;
}

int fsi_create(const FsiMeshDesc* mesh, const FsiParams* prm, int device, FsiCtx** out) {
  FsiTuning t;
  fsi_tuning_from_env(&t);        // defaults + the FSI_<NAME> overrides of the environment (csrc/fsi_tuning.hip)
  return fsi_create_tuned(mesh, prm, device, &t, out);
}

int fsi_get_tuning(const FsiCtx* ctx, FsiTuning* out) {
  if (!ctx || !out) return FSI_ERR_INVALID;
  fsi_tuning_copy_out(&ctx->tune, out);      // out->struct_size is in / out: at most that many bytes are written
  return FSI_OK;
}

// FsiTuning -> the context's fields (the kernels and the solver read those)
static void apply_tuning(FsiCtx* ctx, const FsiTuning& t) {
  ctx->tune = t;
  ctx->kry_fp32_policy = t.krylov_fp32; ctx->op32_policy = t.operator_fp32; ctx->schur_fp32 = t.schur_fp32 != 0;
  ctx->sweeps_fp32 = t.sweeps_fp32; ctx->solid_fp32 = t.solid_fp32;
  ctx->fused_sweeps = t.fused_sweeps != 0; ctx->sweeps_fp16 = ctx->fused_sweeps && t.sweeps_fp16 != 0;
  ctx->coloured = t.node_order == 2;
  ctx->newton_forcing = t.newton_forcing; ctx->newton_forcing_late = t.newton_forcing_late; ctx->newton_late_factor = t.newton_late_factor;
  ctx->f32_cycle_floor = t.f32_cycle_floor; ctx->f32_verdict_skip_rtol = t.f32_verdict_skip_rtol;
  ctx->orth_floor32 = t.orth_floor32; ctx->orth_floor64 = t.orth_floor64; ctx->gcr_escape = t.gcr_escape; ctx->gcr_reorth = t.gcr_reorth;
  ctx->prec_streams = t.prec_streams; ctx->cheb4 = t.cheb4; ctx->coarse_power = t.coarse_power; ctx->solid_mg = t.solid_mg; ctx->dd_mg = t.dd_mg;
  ctx->solid_block_jacobi = t.solid_block_jacobi; ctx->solid_fused = t.solid_fused;
  ctx->cheb_its_s = t.its_solid; ctx->cheb_its_f = t.its_fluid; ctx->cheb_its_p = t.its_schur; ctx->cheb_its_d = t.its_disp;
  ctx->cheb_kappa_s = t.kappa_solid; ctx->cheb_kappa_f = t.kappa_fluid; ctx->cheb_kappa_p = t.kappa_schur; ctx->cheb_kappa_d = t.kappa_disp;
  ctx->solid_coarse_exact = t.solid_coarse_exact;
  ctx->newton_adaptive = t.newton_adaptive;
  ctx->sbmg_pre = t.sbmg_pre; ctx->sbmg_post = t.sbmg_post; ctx->sbmg_cits = t.sbmg_cits; ctx->sbmg_alpha = t.sbmg_alpha; ctx->sbmg_ckappa = t.sbmg_ckappa;
  ctx->mg_pre = t.mg_pre; ctx->mg_post = t.mg_post; ctx->mg_cits = t.mg_cits; ctx->mg_alpha = t.mg_alpha; ctx->mg_ckappa = t.mg_ckappa;
}

int fsi_create_tuned(const FsiMeshDesc* mesh, const FsiParams* prm, int device, const FsiTuning* tuning, FsiCtx** out) {
  if (!mesh || !prm || !out) return FSI_ERR_INVALID;
  *out = nullptr;
  FsiCtx* ctx = new FsiCtx();
  *out = ctx;   // returned even on failure so that fsi_last_error() can be read; caller destroys it
  {
    FsiTuning t;
    fsi_tuning_defaults(&t);
    if (tuning) {      // a caller built against a shorter struct: its fields, the defaults for the rest
      const size_t n = std::min<size_t>(sizeof(FsiTuning), tuning->struct_size > 0 ? (size_t)tuning->struct_size : sizeof(FsiTuning));
      std::memcpy(&t, tuning, n);
      t.struct_size = (int32_t)sizeof(FsiTuning);
    }
    if (t.its_schur <= 0 || t.its_disp <= 0 || t.its_fluid <= 0 || t.its_solid <= 0 || t.krylov_capacity < 8 || t.krylov_fp32 < 0 || t.krylov_fp32 > 2 ||
        (t.jacobian_waves != 1 && t.jacobian_waves != 2) || !(t.newton_forcing >= 0.0)) {
      ctx->err = "fsi_create: FsiTuning out of range (sweep counts must be positive, krylov_capacity >= 8, krylov_fp32 in 0..2, jacobian_waves 1 | 2)";
      return FSI_ERR_INVALID;
    }
    apply_tuning(ctx, t);
  }
  ctx->device = device;
  const int64_t V = mesh->num_vertices, N2 = mesh->num_nodes, C = mesh->num_cells;
  if (V <= 0 || N2 < V || C <= 0 || !mesh->coords || !mesh->tet_nodes || !mesh->cell_kind || !mesh->cell_region) {
    ctx->err = "fsi_create: empty or inconsistent mesh description";
    return FSI_ERR_INVALID;
  }
  if (prm->num_fluid_regions > MAX_REGIONS || prm->num_solid_regions > MAX_REGIONS || !(prm->dt > 0.0)) {
    ctx->err = "fsi_create: bad parameters (dt <= 0 or too many regions)";
    return FSI_ERR_INVALID;
  }
  if (6 * N2 + V >= (int64_t)2147483647) { ctx->err = "fsi_create: more than 2^31 dofs"; return FSI_ERR_INVALID; }
  for (int64_t c = 0; c < C; ++c) {
    const int kind = mesh->cell_kind[c], reg = mesh->cell_region[c];
    if (kind < 0 || kind > 1 || reg < 0 || reg >= (kind == 0 ? prm->num_fluid_regions : prm->num_solid_regions)) {
      ctx->err = "fsi_create: cell with a bad kind/region marker";
      return FSI_ERR_INVALID;
    }
    for (int a = 0; a < 10; ++a) {
      const int32_t nd = mesh->tet_nodes[10 * c + a];
      if (nd < 0 || nd >= N2 || (a < 4 && nd >= V)) { ctx->err = "fsi_create: node id out of range"; return FSI_ERR_INVALID; }
    }
  }
  for (int r = 0; r < prm->num_solid_regions; ++r)
    if (prm->solid_models && (prm->solid_models[r] < 0 || prm->solid_models[r] > 1)) { ctx->err = "fsi_create: material model must be 0 (StVenantKirchoff) or 1 (MooneyRivlin)"; return FSI_ERR_INVALID; }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) { ctx->err = "fsi_create: no such HIP device"; return FSI_ERR_DEVICE; }
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreate(&ctx->stream));
  HIPCHK(hipStreamCreate(&ctx->stream2));
  for (hipEvent_t* e : {&ctx->ev_split, &ctx->ev_solid, &ctx->ev_b}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  HIPCHK(hipEventCreate(&ctx->ev0));
  HIPCHK(hipEventCreate(&ctx->ev1));
  HIPCHK(upload_tables());

  ctx->V = V; ctx->N2 = N2; ctx->C = C; ctx->ndof = 6 * N2 + V;
  if (ctx->tune.mg_post <= 0) {
    // "by size" (the default): what a fine displacement sweep costs is a launch's latency in a small context and its bytes in a large
    // one, what it saves in outer iterations is the same - seven sweeps after the coarse correction win 2 - 15 % up to 630 k tets in
    // both storage modes, five win 3 - 6 % from 1.12 M tets on in the mixed mode (profiles/r05_param_scan_final_tree.txt)
    ctx->tune.mg_post = N2 < 1100000 ? 7 : 5;
    ctx->mg_post = ctx->tune.mg_post;
  }
  ctx->scheme = Scheme{prm->dt, prm->theta, 1.0 - prm->theta, prm->delta, prm->laplace_alpha};
  ctx->nfluid = prm->num_fluid_regions;
  ctx->nsolid = prm->num_solid_regions;
  for (int r = 0; r < MAX_REGIONS; ++r) { ctx->fluid[r] = FluidProps{1.0, 1.0}; ctx->solid[r] = SolidProps{1.0, 1.0, 1.0, 0, 0.0, 0.0, 0.0}; }
  for (int r = 0; r < ctx->nfluid; ++r) ctx->fluid[r] = FluidProps{prm->fluid_props[2 * r], prm->fluid_props[2 * r + 1]};
  for (int r = 0; r < ctx->nsolid; ++r)
    ctx->solid[r] = SolidProps{prm->solid_props[6 * r], prm->solid_props[6 * r + 1], prm->solid_props[6 * r + 2],
                               prm->solid_models ? prm->solid_models[r] : 0, prm->solid_props[6 * r + 3],
                               prm->solid_props[6 * r + 4], prm->solid_props[6 * r + 5]};
  ctx->h_coords.assign(mesh->coords, mesh->coords + 3 * V);
  ctx->h_tet_nodes.assign(mesh->tet_nodes, mesh->tet_nodes + 10 * C);
  const int32_t* tn = ctx->h_tet_nodes.data();

  // ---- base ordering: every vertex followed by the edge nodes it owns (= edges whose lower vertex it is) ------
  std::vector<int32_t> owner(N2, -1), other(N2, 0);
  for (int32_t v = 0; v < V; ++v) owner[v] = v;
  for (int64_t c = 0; c < C; ++c)
    for (int e = 0; e < 6; ++e) {
      const int32_t a = tn[10 * c + TET_EDGES[e][0]], b = tn[10 * c + TET_EDGES[e][1]], nd = tn[10 * c + 4 + e];
      owner[nd] = std::min(a, b);
      other[nd] = std::max(a, b);
    }
  for (int64_t i = 0; i < N2; ++i)
    if (owner[i] < 0) { ctx->err = "fsi_create: P2 node that belongs to no cell"; return FSI_ERR_INVALID; }
  std::vector<int32_t> base(N2), base_rank(N2);
  std::iota(base.begin(), base.end(), 0);
  std::sort(base.begin(), base.end(), [&](int32_t x, int32_t y) {
    if (owner[x] != owner[y]) return owner[x] < owner[y];
    const bool ex = x >= V, ey = y >= V;
    if (ex != ey) return !ex;
    if (other[x] != other[y]) return other[x] < other[y];
    return x < y;
  });
  {   // default numbering: P2 nodes along a Morton (Z-order) curve through their coordinates - spatially compact runs of
      // consecutive nodes are what the gathers of every SpMV, the element scatters and the LDS tiles live on
    const bool morton = ctx->tune.node_order == 0;      // mesh / multicolour order keep `base`
    if (morton) {
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (int64_t v = 0; v < V; ++v)
        for (int i = 0; i < 3; ++i) { lo[i] = std::min(lo[i], mesh->coords[3 * v + i]); hi[i] = std::max(hi[i], mesh->coords[3 * v + i]); }
      double span = 0.0;
      for (int i = 0; i < 3; ++i) span = std::max(span, hi[i] - lo[i]);
      if (!(span > 0.0)) span = 1.0;
      auto spread = [](uint64_t x) {          // 21 bits -> every third bit
        x &= 0x1fffff;
        x = (x | x << 32) & 0x1f00000000ffffull;
        x = (x | x << 16) & 0x1f0000ff0000ffull;
        x = (x | x << 8) & 0x100f00f00f00f00full;
        x = (x | x << 4) & 0x10c30c30c30c30c3ull;
        x = (x | x << 2) & 0x1249249249249249ull;
        return x;
      };
      std::vector<uint64_t> code(N2);
      for (int64_t nd = 0; nd < N2; ++nd) {
        uint64_t c = 0;
        for (int i = 0; i < 3; ++i) {
          const double x = nd < V ? mesh->coords[3 * nd + i]
                                  : 0.5 * (mesh->coords[3 * (int64_t)owner[nd] + i] + mesh->coords[3 * (int64_t)other[nd] + i]);
          const uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, (x - lo[i]) / span * 2097151.0));
          c |= spread(q) << i;
        }
        code[nd] = c;
      }
      std::sort(base.begin(), base.end(), [&](int32_t a, int32_t b) { return code[a] != code[b] ? code[a] < code[b] : a < b; });
    }
  }
  for (int64_t r = 0; r < N2; ++r) base_rank[base[r]] = (int32_t)r;

  // ---- node graph (node ids), sorted unique pairs ---------------------------------------------------------------
  std::vector<uint64_t> pairs;
  pairs.reserve((size_t)C * 100);
  for (int64_t c = 0; c < C; ++c)
    for (int a = 0; a < 10; ++a)
      for (int b = 0; b < 10; ++b)
        pairs.push_back(((uint64_t)(uint32_t)tn[10 * c + a] << 32) | (uint32_t)tn[10 * c + b]);
  std::sort(pairs.begin(), pairs.end());
  pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
  std::vector<int64_t> gptr(N2 + 1, 0);
  std::vector<int32_t> gadj(pairs.size());
  for (size_t i = 0; i < pairs.size(); ++i) {
    gptr[(pairs[i] >> 32) + 1] += 1;
    gadj[i] = (int32_t)(pairs[i] & 0xffffffffu);
  }
  for (int64_t r = 0; r < N2; ++r) gptr[r + 1] += gptr[r];
  std::vector<uint64_t>().swap(pairs);

  // ---- greedy multicolouring of the node graph (base order): nodes of one colour share no element -----------------
  // (only the ILU(0) paths need it; the default Chebyshev-based preconditioner keeps the mesh's own node order, whose
  //  locality is what the gathers of every SpMV live on.  FSI_ORDER=colour selects the multicolour ordering.)
  ctx->coloured = ctx->tune.node_order == 2;
  std::vector<int32_t> color(N2, ctx->coloured ? -1 : 0);
  if (!ctx->coloured) ctx->ncolors = 1;
  if (ctx->coloured) {
    std::vector<int32_t> mark(1024, -1);
    for (int64_t r = 0; r < N2; ++r) {
      const int32_t nd = base[r];
      for (int64_t k = gptr[nd]; k < gptr[nd + 1]; ++k) {
        const int32_t c = color[gadj[k]];
        if (c >= 0) {
          if ((size_t)c >= mark.size()) mark.resize(2 * c + 2, -1);
          mark[c] = nd;
        }
      }
      int32_t c = 0;
      while ((size_t)c < mark.size() && mark[c] == nd) ++c;
      if ((size_t)c >= mark.size()) mark.resize(2 * c + 2, -1);
      color[nd] = c;
      ctx->ncolors = std::max(ctx->ncolors, c + 1);
    }
  }
  ctx->h_rank2node = base;
  std::stable_sort(ctx->h_rank2node.begin(), ctx->h_rank2node.end(), [&](int32_t x, int32_t y) { return color[x] < color[y]; });
  ctx->h_node2rank.resize(N2);
  for (int64_t r = 0; r < N2; ++r) ctx->h_node2rank[ctx->h_rank2node[r]] = (int32_t)r;
  const std::vector<int32_t>& rk = ctx->h_node2rank;
  // pressure block: vertices in the same (colour, base) order
  ctx->h_prank.assign(V, 0);
  std::vector<int32_t> prow_rank(V);
  {
    int32_t q = 0;
    for (int64_t r = 0; r < N2; ++r) {
      const int32_t nd = ctx->h_rank2node[r];
      if (nd < V) { ctx->h_prank[nd] = q; prow_rank[q] = (int32_t)r; ++q; }
    }
  }
  // levels: one per colour for the d/v rows (6 rows per node), then one per colour for the pressure rows
  {
    std::vector<int64_t> ncount(ctx->ncolors, 0), vcount(ctx->ncolors, 0);
    for (int64_t nd = 0; nd < N2; ++nd) { ncount[color[nd]] += 1; if (nd < V) vcount[color[nd]] += 1; }
    int64_t r0 = 0;
    for (int c = 0; c < ctx->ncolors; ++c) { ctx->levels.push_back(Level{6 * r0, ncount[c], 6}); r0 += ncount[c]; }
    int64_t q0 = 0;
    for (int c = 0; c < ctx->ncolors; ++c) { ctx->levels.push_back(Level{6 * N2 + q0, vcount[c], 1}); q0 += vcount[c]; }
  }
  // adjacency in final ranks (ascending) and pressure neighbours as positions in the pressure block (ascending)
  ctx->h_nadj_ptr.assign(N2 + 1, 0);
  ctx->h_nadj.resize(gadj.size());
  ctx->h_padj_ptr.assign(N2 + 1, 0);
  ctx->h_padj.clear();
  {
    int64_t o = 0;
    std::vector<int32_t> tmpv;
    for (int64_t r = 0; r < N2; ++r) {
      const int32_t nd = ctx->h_rank2node[r];
      const int64_t o0 = o;
      tmpv.clear();
      for (int64_t k = gptr[nd]; k < gptr[nd + 1]; ++k) {
        ctx->h_nadj[o++] = rk[gadj[k]];
        if (gadj[k] < V) tmpv.push_back(ctx->h_prank[gadj[k]]);
      }
      std::sort(ctx->h_nadj.begin() + o0, ctx->h_nadj.begin() + o);
      std::sort(tmpv.begin(), tmpv.end());
      ctx->h_padj.insert(ctx->h_padj.end(), tmpv.begin(), tmpv.end());
      ctx->h_nadj_ptr[r + 1] = o;
      ctx->h_padj_ptr[r + 1] = (int64_t)ctx->h_padj.size();
    }
  }
  std::vector<int64_t>().swap(gptr);
  std::vector<int32_t>().swap(gadj);
  for (int64_t r = 0; r < N2; ++r) {
    const int64_t deg = ctx->h_nadj_ptr[r + 1] - ctx->h_nadj_ptr[r];
    if (deg >= 65536 / 6 || 6 * deg + (ctx->h_padj_ptr[r + 1] - ctx->h_padj_ptr[r]) > 1024) {
      ctx->err = "fsi_create: node with too many neighbours for the row buffers (max 1024 entries per row)";
      return FSI_ERR_INVALID;
    }
  }

  // ---- CSR row pointers ------------------------------------------------------------------------------------
  std::vector<int64_t> rowptr(ctx->ndof + 1, 0);
  for (int64_t r = 0; r < N2; ++r) {
    const int64_t len = 6 * (ctx->h_nadj_ptr[r + 1] - ctx->h_nadj_ptr[r]) + (ctx->h_padj_ptr[r + 1] - ctx->h_padj_ptr[r]);
    for (int t = 0; t < 6; ++t) rowptr[6 * r + t + 1] = len;
  }
  for (int64_t q = 0; q < V; ++q) {
    const int32_t r = prow_rank[q];
    rowptr[6 * N2 + q + 1] = 6 * (ctx->h_nadj_ptr[r + 1] - ctx->h_nadj_ptr[r]) + (ctx->h_padj_ptr[r + 1] - ctx->h_padj_ptr[r]);
  }
  for (int64_t i = 0; i < ctx->ndof; ++i) rowptr[i + 1] += rowptr[i];
  ctx->nnz = rowptr[ctx->ndof];

  // ---- element tables --------------------------------------------------------------------------------------
  std::vector<int32_t> cell_dofs((size_t)C * NLOC), cell_rank((size_t)C * 10), tet_vertices((size_t)C * 4), cell_prow((size_t)C * 4);
  std::vector<uint16_t> enbr((size_t)C * 100), epnbr((size_t)C * 40);
  for (int64_t c = 0; c < C; ++c) {
    for (int a = 0; a < 10; ++a) {
      const int32_t ra = rk[tn[10 * c + a]];
      cell_rank[10 * c + a] = ra;
      for (int cmp = 0; cmp < 3; ++cmp) {
        cell_dofs[c * NLOC + cmp * 10 + a] = 6 * ra + cmp;
        cell_dofs[c * NLOC + 30 + cmp * 10 + a] = 6 * ra + 3 + cmp;
      }
      const int32_t* lo = ctx->h_nadj.data() + ctx->h_nadj_ptr[ra];
      const int32_t* hi = ctx->h_nadj.data() + ctx->h_nadj_ptr[ra + 1];
      for (int b = 0; b < 10; ++b)
        enbr[c * 100 + a * 10 + b] = (uint16_t)(std::lower_bound(lo, hi, rk[tn[10 * c + b]]) - lo);
      const int32_t* plo = ctx->h_padj.data() + ctx->h_padj_ptr[ra];
      const int32_t* phi = ctx->h_padj.data() + ctx->h_padj_ptr[ra + 1];
      for (int b = 0; b < 4; ++b)
        epnbr[c * 40 + a * 4 + b] = (uint16_t)(std::lower_bound(plo, phi, ctx->h_prank[tn[10 * c + b]]) - plo);
    }
    for (int a = 0; a < 4; ++a) {
      cell_dofs[c * NLOC + 60 + a] = (int32_t)(6 * N2 + ctx->h_prank[tn[10 * c + a]]);
      cell_prow[4 * c + a] = cell_dofs[c * NLOC + 60 + a];
      tet_vertices[4 * c + a] = tn[10 * c + a];
    }
  }
  // ---- assembly colouring: greedy, balanced (the least used admissible colour), at most 128 colours ---------------------
  ctx->ncellcol = 0;
  {
    if (!ctx->tune.assembly_atomic && C > 0) {
      constexpr int MAXCOL = 128;
      std::vector<uint64_t> used((size_t)N2 * 2, 0);
      std::vector<uint8_t> colour((size_t)C);
      std::vector<int64_t> count;
      bool ok = true;
      for (int64_t c = 0; c < C && ok; ++c) {
        uint64_t m0 = 0, m1 = 0;
        for (int a = 0; a < 10; ++a) { m0 |= used[2 * (size_t)cell_rank[10 * c + a]]; m1 |= used[2 * (size_t)cell_rank[10 * c + a] + 1]; }
        int best = -1;
        for (int k = 0; k < (int)count.size(); ++k) {
          const bool taken = k < 64 ? (m0 >> k) & 1 : (m1 >> (k - 64)) & 1;
          if (!taken && (best < 0 || count[k] < count[best])) best = k;
        }
        if (best < 0) {
          if ((int)count.size() == MAXCOL) { ok = false; break; }
          best = (int)count.size();
          count.push_back(0);
        }
        colour[c] = (uint8_t)best;
        count[best] += 1;
        for (int a = 0; a < 10; ++a) {
          if (best < 64) used[2 * (size_t)cell_rank[10 * c + a]] |= 1ull << best;
          else used[2 * (size_t)cell_rank[10 * c + a] + 1] |= 1ull << (best - 64);
        }
      }
      if (ok) {
        const int nc = (int)count.size();
        ctx->h_col_ptr.assign(nc + 1, 0);
        for (int k = 0; k < nc; ++k) ctx->h_col_ptr[k + 1] = ctx->h_col_ptr[k] + count[k];
        std::vector<int64_t> fill(ctx->h_col_ptr.begin(), ctx->h_col_ptr.end() - 1);
        std::vector<int32_t> cells((size_t)C);
        for (int64_t c = 0; c < C; ++c) cells[fill[colour[c]]++] = (int32_t)c;
        FSICHK(upload(ctx, ctx->col_cells, cells));
        ctx->ncellcol = nc;
      }   // more than 128 cells around one node: the unordered single launch stays (ncellcol = 0)
      // incidences of the residual gather: per node rank / pressure row the (cell, local index) pairs, cells ascending
      if (C < (int64_t)1 << 27) {
        std::vector<int64_t> iptr((size_t)N2 + 1, 0), pptr((size_t)V + 1, 0);
        for (int64_t c = 0; c < C; ++c) {
          for (int a = 0; a < 10; ++a) iptr[(size_t)cell_rank[10 * c + a] + 1] += 1;
          for (int a = 0; a < 4; ++a) pptr[(size_t)(cell_prow[4 * c + a] - 6 * N2) + 1] += 1;
        }
        for (int64_t r = 0; r < N2; ++r) iptr[r + 1] += iptr[r];
        for (int64_t q = 0; q < V; ++q) pptr[q + 1] += pptr[q];
        std::vector<int32_t> inc((size_t)10 * C), pinc((size_t)4 * C);
        std::vector<int64_t> ifill(iptr.begin(), iptr.end() - 1), pfill(pptr.begin(), pptr.end() - 1);
        for (int64_t c = 0; c < C; ++c) {
          for (int a = 0; a < 10; ++a) inc[ifill[cell_rank[10 * c + a]]++] = (int32_t)(16 * c + a);
          for (int a = 0; a < 4; ++a) pinc[pfill[cell_prow[4 * c + a] - 6 * N2]++] = (int32_t)(16 * c + a);
        }
        FSICHK(upload(ctx, ctx->inc_ptr, iptr));
        FSICHK(upload(ctx, ctx->pinc_ptr, pptr));
        FSICHK(upload(ctx, ctx->inc, inc));
        FSICHK(upload(ctx, ctx->pinc, pinc));
        HIPCHK(ctx->Re.alloc((size_t)C * NLOC));
      }
    }
  }
  ctx->h_user2solver.resize(ctx->ndof);
  std::vector<int32_t> solver2user(ctx->ndof);
  for (int64_t nd = 0; nd < N2; ++nd)
    for (int cmp = 0; cmp < 3; ++cmp) {
      ctx->h_user2solver[3 * nd + cmp] = 6 * rk[nd] + cmp;
      ctx->h_user2solver[3 * N2 + 3 * nd + cmp] = 6 * rk[nd] + 3 + cmp;
    }
  for (int64_t v = 0; v < V; ++v) ctx->h_user2solver[6 * N2 + v] = (int32_t)(6 * N2 + ctx->h_prank[v]);
  for (int64_t i = 0; i < ctx->ndof; ++i) solver2user[ctx->h_user2solver[i]] = (int32_t)i;

  // ---- upload ------------------------------------------------------------------------------------------------
  FSICHK(upload(ctx, ctx->user2solver, ctx->h_user2solver));
  FSICHK(upload(ctx, ctx->solver2user, solver2user));
  FSICHK(upload(ctx, ctx->cell_dofs, cell_dofs));
  FSICHK(upload(ctx, ctx->cell_rank, cell_rank));
  FSICHK(upload(ctx, ctx->cell_prow, cell_prow));
  FSICHK(upload(ctx, ctx->enbr, enbr));
  FSICHK(upload(ctx, ctx->epnbr, epnbr));
  FSICHK(upload(ctx, ctx->cell_kind, std::vector<int32_t>(mesh->cell_kind, mesh->cell_kind + C)));
  FSICHK(upload(ctx, ctx->cell_region, std::vector<int32_t>(mesh->cell_region, mesh->cell_region + C)));
  FSICHK(upload(ctx, ctx->nadj_ptr, ctx->h_nadj_ptr));
  FSICHK(upload(ctx, ctx->nadj, ctx->h_nadj));
  FSICHK(upload(ctx, ctx->padj_ptr, ctx->h_padj_ptr));
  FSICHK(upload(ctx, ctx->padj, ctx->h_padj));
  FSICHK(upload(ctx, ctx->rowptr, rowptr));
  HIPCHK(ctx->cols.alloc(ctx->nnz));
  HIPCHK(ctx->diagpos.alloc(ctx->ndof));
  FSICHK(upload(ctx, ctx->vrank, prow_rank));
  {
    DevBuf<int32_t>& d_vrank = ctx->vrank;
    DevBuf<int32_t> d_tv;
    DevBuf<double> d_coords;
    FSICHK(upload(ctx, d_tv, tet_vertices));
    FSICHK(upload(ctx, d_coords, ctx->h_coords));
    HIPCHK(ctx->geom.alloc((size_t)C * 10));
    launch_expand_cols(ctx->stream, N2, V, ctx->nadj_ptr.p, ctx->nadj.p, ctx->padj_ptr.p, ctx->padj.p, d_vrank.p,
                       ctx->rowptr.p, ctx->cols.p, ctx->diagpos.p);
    launch_geometry(ctx->stream, C, d_coords.p, d_tv.p, ctx->geom.p);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    d_tv.release(); d_coords.release();
  }
  const int64_t n = ctx->ndof;
  DevBuf<double>* vecs[] = {&ctx->U, &ctx->U1, &ctx->F, &ctx->b, &ctx->du, &ctx->bs, &ctx->tmp1, &ctx->tmp2, &ctx->tmp3,
                            &ctx->tmp4, &ctx->tmp5, &ctx->tmp6, &ctx->tmp7, &ctx->rowscale};
  for (auto* v : vecs) {
    HIPCHK(v->alloc(n));
    HIPCHK(hipMemsetAsync(v->p, 0, n * sizeof(double), ctx->stream));
  }
  HIPCHK(ctx->A_pre.alloc(ctx->nnz));
  HIPCHK(ctx->A.alloc(ctx->nnz));
  // ---- field blocks of the block preconditioner ------------------------------------------------------------------
  {
    std::vector<int32_t> node_solid(N2, 0);
    for (int64_t c = 0; c < C; ++c)
      if (mesh->cell_kind[c] == 1)
        for (int a = 0; a < 10; ++a) node_solid[rk[tn[10 * c + a]]] = 1;
    FSICHK(upload(ctx, ctx->node_solid, node_solid));
    {
      std::vector<double> ms(3 * N2), mf(3 * N2);
      for (int64_t r = 0; r < N2; ++r)
        for (int i = 0; i < 3; ++i) { ms[3 * r + i] = node_solid[r] ? 1.0 : 0.0; mf[3 * r + i] = node_solid[r] ? 0.0 : 1.0; }
      std::vector<int32_t> snode, sidx(N2, -1);
      for (int64_t r = 0; r < N2; ++r)
        if (node_solid[r]) { sidx[r] = (int32_t)snode.size(); snode.push_back((int32_t)r); }
      const int64_t nS = (int64_t)snode.size();
      ctx->nS = nS;
      std::vector<int64_t> ss_rowptr(3 * nS + 1, 0), ss_diagpos(3 * nS, 0), ss_src;
      std::vector<int32_t> ss_cols;
      for (int64_t i = 0; i < nS; ++i) {
        const int64_t r = snode[i], a = ctx->h_nadj_ptr[r], deg = ctx->h_nadj_ptr[r + 1] - a;
        for (int c = 0; c < 3; ++c) {
          const int64_t row0 = 9 * a + 3 * c * deg;            // start of row 3r+c in the 3x3-blocked structure
          for (int64_t k = 0; k < deg; ++k) {
            const int32_t si = sidx[ctx->h_nadj[a + k]];
            if (si < 0) continue;
            for (int j = 0; j < 3; ++j) {
              if (si == i && j == c) ss_diagpos[3 * i + c] = (int64_t)ss_cols.size();
              ss_cols.push_back(3 * si + j);
              ss_src.push_back(row0 + 3 * k + j);
            }
          }
          ss_rowptr[3 * i + c + 1] = (int64_t)ss_cols.size();
        }
      }
      {   // block-CSR structure of the same block
        std::vector<int64_t> sb_ptr(nS + 1, 0), sb_src;
        std::vector<int32_t> sb_col, sb_row, sb_stride(nS);
        for (int64_t i = 0; i < nS; ++i) {
          const int64_t r = snode[i], a = ctx->h_nadj_ptr[r], deg = ctx->h_nadj_ptr[r + 1] - a;
          sb_stride[i] = (int32_t)(3 * deg);
          for (int64_t k = 0; k < deg; ++k) {
            const int32_t si = sidx[ctx->h_nadj[a + k]];
            if (si < 0) continue;
            sb_col.push_back(si);
            sb_row.push_back((int32_t)i);
            sb_src.push_back(9 * a + 3 * k);
          }
          sb_ptr[i + 1] = (int64_t)sb_col.size();
        }
        ctx->sb_nblocks = (int64_t)sb_col.size();
        ctx->h_sb_ptr = sb_ptr;
        ctx->h_sb_col = sb_col;
        ctx->h_snode = snode;
        FSICHK(upload(ctx, ctx->sb_ptr, sb_ptr));
        FSICHK(upload(ctx, ctx->sb_src, sb_src));
        FSICHK(upload(ctx, ctx->sb_col, sb_col));
        FSICHK(upload(ctx, ctx->sb_row, sb_row));
        FSICHK(upload(ctx, ctx->sb_stride, sb_stride));
        HIPCHK(ctx->sb_vals.alloc(9 * sb_col.size()));
        HIPCHK(ctx->sb_dinv.alloc(4 * nS));
        HIPCHK(ctx->sb_binv12.alloc(12 * nS));
        HIPCHK(ctx->sb_binv9.alloc(9 * nS));
      }
      {   // rows of fluid-interior nodes that see solid columns: the only rows the solid predictor changes in the fluid rhs
        std::vector<int32_t> fs_rows, fs_col;
        std::vector<int64_t> fs_ptr(1, 0), fs_src;
        for (int64_t r = 0; r < N2; ++r) {
          if (node_solid[r]) continue;
          const int64_t a = ctx->h_nadj_ptr[r], deg = ctx->h_nadj_ptr[r + 1] - a;
          bool any = false;
          for (int64_t k = 0; k < deg && !any; ++k) any = node_solid[ctx->h_nadj[a + k]] != 0;
          if (!any) continue;
          for (int c = 0; c < 3; ++c) {
            const int64_t row0 = 9 * a + 3 * c * deg;
            for (int64_t k = 0; k < deg; ++k) {
              const int32_t nb = ctx->h_nadj[a + k];
              if (!node_solid[nb]) continue;
              for (int j = 0; j < 3; ++j) { fs_col.push_back(3 * nb + j); fs_src.push_back(row0 + 3 * k + j); }
            }
            fs_rows.push_back((int32_t)(3 * r + c));
            fs_ptr.push_back((int64_t)fs_col.size());
          }
        }
        ctx->nfs = (int64_t)fs_rows.size();
        FSICHK(upload(ctx, ctx->fs_rows, fs_rows));
        FSICHK(upload(ctx, ctx->fs_ptr, fs_ptr));
        FSICHK(upload(ctx, ctx->fs_col, fs_col));
        FSICHK(upload(ctx, ctx->fs_src, fs_src));
      }
      FSICHK(upload(ctx, ctx->snode, snode));
      FSICHK(upload(ctx, ctx->ss_rowptr, ss_rowptr));
      FSICHK(upload(ctx, ctx->ss_diagpos, ss_diagpos));
      FSICHK(upload(ctx, ctx->ss_cols, ss_cols));
      FSICHK(upload(ctx, ctx->ss_src, ss_src));
      HIPCHK(ctx->ss_vals.alloc(ss_cols.size()));
      for (int k = 0; k < 8; ++k) { HIPCHK(hipEventCreate(&ctx->ss_ev0[k])); HIPCHK(hipEventCreate(&ctx->ss_ev1[k])); }
      for (int k = 0; k < 8; ++k) { HIPCHK(hipEventCreate(&ctx->db_ev0[k])); HIPCHK(hipEventCreate(&ctx->db_ev1[k])); }
      for (int k = 0; k < 4; ++k) { HIPCHK(hipEventCreate(&ctx->sc_ev0[k])); HIPCHK(hipEventCreate(&ctx->sc_ev1[k])); }
      for (int k = 0; k < 4; ++k) { HIPCHK(hipEventCreate(&ctx->sch_ev0[k])); HIPCHK(hipEventCreate(&ctx->sch_ev1[k])); }
      FSICHK(upload(ctx, ctx->mask_s, ms));
      FSICHK(upload(ctx, ctx->mask_f, mf));
    }
    const int64_t nadj_total = ctx->h_nadj_ptr[N2], padj_total = ctx->h_padj_ptr[N2];
    std::vector<int64_t> rowptr_pv(V + 1, 0), rowptr_pp(V + 1, 0), diagpos_pp(V, 0);
    for (int64_t q = 0; q < V; ++q) {
      const int32_t r = prow_rank[q];
      rowptr_pv[q + 1] = rowptr_pv[q] + 3 * (ctx->h_nadj_ptr[r + 1] - ctx->h_nadj_ptr[r]);
      rowptr_pp[q + 1] = rowptr_pp[q] + (ctx->h_padj_ptr[r + 1] - ctx->h_padj_ptr[r]);
    }
    std::vector<int32_t> cols_pp(rowptr_pp[V]);
    for (int64_t q = 0; q < V; ++q) {
      const int32_t r = prow_rank[q];
      const int64_t a = ctx->h_padj_ptr[r], len = ctx->h_padj_ptr[r + 1] - a;
      bool found = false;
      for (int64_t k = 0; k < len; ++k) {
        cols_pp[rowptr_pp[q] + k] = ctx->h_padj[a + k];
        if (ctx->h_padj[a + k] == q) { diagpos_pp[q] = rowptr_pp[q] + k; found = true; }
      }
      if (!found) { ctx->err = "fsi_create: vertex missing from its own neighbour list"; return FSI_ERR_INVALID; }
    }
    FSICHK(upload(ctx, ctx->rowptr_pv, rowptr_pv));
    FSICHK(upload(ctx, ctx->rowptr_pp, rowptr_pp));
    FSICHK(upload(ctx, ctx->diagpos_pp, diagpos_pp));
    FSICHK(upload(ctx, ctx->cols_pp, cols_pp));
    HIPCHK(ctx->rowptr3.alloc(3 * N2 + 1));
    HIPCHK(ctx->diagpos3.alloc(3 * N2));
    HIPCHK(ctx->cols3.alloc(9 * nadj_total));
    HIPCHK(ctx->rowptr_vp.alloc(3 * N2 + 1));
    HIPCHK(ctx->cols_vp.alloc(3 * padj_total));
    HIPCHK(ctx->cols_pv.alloc(rowptr_pv[V]));
    launch_block_structure(ctx->stream, N2, V, ctx->nadj_ptr.p, ctx->nadj.p, ctx->padj_ptr.p, ctx->padj.p, ctx->vrank.p,
                           ctx->rowptr3.p, ctx->cols3.p, ctx->diagpos3.p, ctx->rowptr_vp.p, ctx->cols_vp.p,
                           ctx->rowptr_pv.p, ctx->cols_pv.p);
    HIPCHK(hipGetLastError());
    HIPCHK(ctx->Adv.alloc(9 * nadj_total));
    HIPCHK(ctx->dd_db.alloc(3 * nadj_total));
    HIPCHK(ctx->vv_db.alloc(3 * nadj_total));
    HIPCHK(ctx->adv_db.alloc(3 * nadj_total));
    HIPCHK(ctx->dd_db32.alloc(3 * nadj_total));
    HIPCHK(ctx->vv_db32.alloc(3 * nadj_total));
    HIPCHK(ctx->dd_dinv32.alloc(4 * N2));
    HIPCHK(ctx->dd_chat.alloc(nadj_total));
    HIPCHK(ctx->dd_rowflag.alloc(3 * N2));
    {   // LDS tiles of the node graph: per tile of consecutive nodes the sorted distinct column nodes + local indices
      // nodes per tile (FsiTuning.tile_nodes; 0 = 256): 128 makes no difference at 140 k tets (49.7 against 49.9 ms per step) and
      // costs 1.5 % at 1.12 M tets (round 4 scan, profiles/r04_tile_scan.txt)
      const int TN = (ctx->tune.tile_nodes == 128 || ctx->tune.tile_nodes == 256) ? ctx->tune.tile_nodes : 256;
      const int LIM = tile_limit();
      ctx->tile_nodes = TN;
      const int64_t ntiles = (N2 + TN - 1) / TN;
      std::vector<int64_t> uptr(ntiles + 1, 0);
      std::vector<int32_t> ulist;
      std::vector<uint16_t> ploc(nadj_total);
      std::vector<int32_t> tmpu;
      bool ok = ctx->tune.tiles != 0;
      for (int64_t t = 0; t < ntiles && ok; ++t) {
        const int64_t r0 = t * TN, r1 = std::min<int64_t>(N2, r0 + TN);
        const int64_t e0 = ctx->h_nadj_ptr[r0], e1 = ctx->h_nadj_ptr[r1];
        tmpu.assign(ctx->h_nadj.begin() + e0, ctx->h_nadj.begin() + e1);
        std::sort(tmpu.begin(), tmpu.end());
        tmpu.erase(std::unique(tmpu.begin(), tmpu.end()), tmpu.end());
        if ((int64_t)tmpu.size() > LIM) { ok = false; break; }
        ctx->tile_max_nu = std::max<int>(ctx->tile_max_nu, (int)tmpu.size());
        for (int64_t e = e0; e < e1; ++e)
          ploc[e] = (uint16_t)(std::lower_bound(tmpu.begin(), tmpu.end(), ctx->h_nadj[e]) - tmpu.begin());
        ulist.insert(ulist.end(), tmpu.begin(), tmpu.end());
        uptr[t + 1] = (int64_t)ulist.size();
      }
      ctx->tiled = ok;
      if (ok) {
        FSICHK(upload(ctx, ctx->tile_uptr, uptr));
        FSICHK(upload(ctx, ctx->tile_ulist, ulist));
        FSICHK(upload(ctx, ctx->tile_ploc, ploc));
        std::vector<int> sorted_nu(ntiles);
        for (int64_t t = 0; t < ntiles; ++t) sorted_nu[t] = (int)(uptr[t + 1] - uptr[t]);
        std::sort(sorted_nu.begin(), sorted_nu.end());
        if (getenv("FSI_DEBUG_PRECOND"))
          fprintf(stderr, "[precond] tiles: %lld, distinct neighbours median %d, 90%% %d, max %d\n", (long long)ntiles,
                  sorted_nu[(size_t)(ntiles / 2)], sorted_nu[(size_t)((ntiles - 1) * 9 / 10)], ctx->tile_max_nu);
      }
    }
    {
      std::vector<float> ones(4 * N2, 1.0f);
      for (int64_t i = 0; i < N2; ++i) ones[4 * i + 3] = 0.0f;
      FSICHK(upload(ctx, ctx->ones32, ones));
    }
    {   // P2 -> P1 hierarchy of the displacement block: parents of every node, vertex graph, children of every vertex
      std::vector<int32_t> cidx(N2, -1), cfine;
      for (int64_t r = 0; r < N2; ++r)
        if (ctx->h_rank2node[r] < V) { cidx[r] = (int32_t)cfine.size(); cfine.push_back((int32_t)r); }
      const int64_t nc = (int64_t)cfine.size();
      static const int TE[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};      // UFC edge -> local vertices
      std::vector<int32_t> ends(2 * (size_t)N2, -1);
      for (int64_t c = 0; c < C; ++c)
        for (int e = 0; e < 6; ++e) {
          const int32_t nd = tn[10 * c + 4 + e];
          ends[2 * (size_t)nd] = tn[10 * c + TE[e][0]];
          ends[2 * (size_t)nd + 1] = tn[10 * c + TE[e][1]];
        }
      std::vector<int32_t> par(2 * (size_t)N2);
      std::vector<float> pw(2 * (size_t)N2);
      std::vector<int64_t> chptr(nc + 1, 0);
      bool ok = nc == V;
      for (int64_t r = 0; r < N2 && ok; ++r) {
        const int32_t nd = ctx->h_rank2node[r];
        if (nd < V) { par[2 * r] = par[2 * r + 1] = cidx[r]; pw[2 * r] = 1.f; pw[2 * r + 1] = 0.f; chptr[cidx[r] + 1] += 1; }
        else {
          const int32_t a = ends[2 * (size_t)nd], b = ends[2 * (size_t)nd + 1];
          if (a < 0 || b < 0 || a >= V || b >= V) { ok = false; break; }
          par[2 * r] = cidx[rk[a]]; par[2 * r + 1] = cidx[rk[b]];
          pw[2 * r] = pw[2 * r + 1] = 0.5f;
          chptr[par[2 * r] + 1] += 1; chptr[par[2 * r + 1] + 1] += 1;
        }
      }
      if (ok) {
        for (int64_t i = 0; i < nc; ++i) chptr[i + 1] += chptr[i];
        std::vector<int32_t> child(chptr[nc]);
        std::vector<float> chw(chptr[nc]);
        std::vector<int64_t> fill(chptr.begin(), chptr.end() - 1);
        for (int64_t r = 0; r < N2; ++r)
          for (int k = 0; k < 2; ++k)
            if (pw[2 * r + k] != 0.f) { const int64_t pos = fill[par[2 * r + k]]++; child[pos] = (int32_t)r; chw[pos] = pw[2 * r + k]; }
        std::vector<int64_t> cptr(nc + 1, 0);
        std::vector<int32_t> ccol;
        for (int64_t i = 0; i < nc; ++i) {
          const int64_t r = cfine[i];
          for (int64_t e = ctx->h_nadj_ptr[r]; e < ctx->h_nadj_ptr[r + 1]; ++e)
            if (cidx[ctx->h_nadj[e]] >= 0) ccol.push_back(cidx[ctx->h_nadj[e]]);      // ascending: ranks ascend, cidx is monotone
          cptr[i + 1] = (int64_t)ccol.size();
        }
        ctx->mg_nc = nc;
        ctx->mg_cnnz = (int64_t)ccol.size();
        FSICHK(upload(ctx, ctx->mg_par, par));
        FSICHK(upload(ctx, ctx->mg_pw, pw));
        FSICHK(upload(ctx, ctx->mg_chptr, chptr));
        FSICHK(upload(ctx, ctx->mg_child, child));
        FSICHK(upload(ctx, ctx->mg_chw, chw));
        FSICHK(upload(ctx, ctx->mg_cptr, cptr));
        FSICHK(upload(ctx, ctx->mg_ccol, ccol));
        FSICHK(upload(ctx, ctx->mg_cfine, cfine));
        HIPCHK(ctx->mg_Ac.alloc(ctx->mg_cnnz));
        HIPCHK(ctx->mg_cc.alloc(ctx->mg_cnnz));
        HIPCHK(ctx->mg_d0.alloc(N2));
        HIPCHK(ctx->mg_dcinv4.alloc(4 * nc));
        HIPCHK(ctx->mg_cflag.alloc(3 * nc));
        HIPCHK(ctx->mg_work.alloc(5 * 4 * nc));
        std::vector<float> cones(4 * (size_t)nc, 1.0f);
        for (int64_t i = 0; i < nc; ++i) cones[4 * i + 3] = 0.0f;
        FSICHK(upload(ctx, ctx->mg_cones, cones));
        // the same hierarchy on the compact solid numbering (3x3-block operator of the velocity predictor)
        const int64_t nS = ctx->nS;
        if (ctx->solid_mg && nS > 0) {
          std::vector<int32_t> sidx2(N2, -1), scidx(nS, -1), scfine;
          for (int64_t i = 0; i < nS; ++i) sidx2[ctx->h_snode[i]] = (int32_t)i;
          for (int64_t i = 0; i < nS; ++i)
            if (ctx->h_rank2node[ctx->h_snode[i]] < V) { scidx[i] = (int32_t)scfine.size(); scfine.push_back((int32_t)i); }
          const int64_t nsc = (int64_t)scfine.size();
          std::vector<int32_t> spar(2 * (size_t)nS);
          std::vector<float> spw(2 * (size_t)nS);
          std::vector<int64_t> schptr(nsc + 1, 0);
          bool sok = nsc > 0;
          for (int64_t i = 0; i < nS && sok; ++i) {
            const int32_t nd = ctx->h_rank2node[ctx->h_snode[i]];
            if (nd < V) { spar[2 * i] = spar[2 * i + 1] = scidx[i]; spw[2 * i] = 1.f; spw[2 * i + 1] = 0.f; schptr[scidx[i] + 1] += 1; }
            else {
              const int32_t ia = sidx2[rk[ends[2 * (size_t)nd]]], ib = sidx2[rk[ends[2 * (size_t)nd + 1]]];
              if (ia < 0 || ib < 0 || scidx[ia] < 0 || scidx[ib] < 0) { sok = false; break; }   // an end vertex outside the solid set
              spar[2 * i] = scidx[ia]; spar[2 * i + 1] = scidx[ib];
              spw[2 * i] = spw[2 * i + 1] = 0.5f;
              schptr[scidx[ia] + 1] += 1; schptr[scidx[ib] + 1] += 1;
            }
          }
          if (sok) {
            for (int64_t i = 0; i < nsc; ++i) schptr[i + 1] += schptr[i];
            std::vector<int32_t> schild(schptr[nsc]);
            std::vector<float> schw(schptr[nsc]);
            std::vector<int64_t> sfill(schptr.begin(), schptr.end() - 1);
            for (int64_t i = 0; i < nS; ++i)
              for (int k = 0; k < 2; ++k)
                if (spw[2 * i + k] != 0.f) { const int64_t pos = sfill[spar[2 * i + k]]++; schild[pos] = (int32_t)i; schw[pos] = spw[2 * i + k]; }
            std::vector<int64_t> scptr(nsc + 1, 0);
            std::vector<int32_t> sccol;
            for (int64_t I = 0; I < nsc; ++I) {
              const int64_t i = scfine[I];
              for (int64_t b = ctx->h_sb_ptr[i]; b < ctx->h_sb_ptr[i + 1]; ++b)
                if (scidx[ctx->h_sb_col[b]] >= 0) sccol.push_back(scidx[ctx->h_sb_col[b]]);
              scptr[I + 1] = (int64_t)sccol.size();
            }
            ctx->sbmg_nc = nsc;
            ctx->sbmg_nblk = (int64_t)sccol.size();
            FSICHK(upload(ctx, ctx->sbmg_par, spar));
            FSICHK(upload(ctx, ctx->sbmg_pw, spw));
            FSICHK(upload(ctx, ctx->sbmg_chptr, schptr));
            FSICHK(upload(ctx, ctx->sbmg_child, schild));
            FSICHK(upload(ctx, ctx->sbmg_chw, schw));
            FSICHK(upload(ctx, ctx->sbmg_cptr, scptr));
            FSICHK(upload(ctx, ctx->sbmg_ccol, sccol));
            FSICHK(upload(ctx, ctx->sbmg_cfine, scfine));
            HIPCHK(ctx->sbmg_cvals.alloc(9 * sccol.size()));
            HIPCHK(ctx->sbmg_cbinv12.alloc(12 * nsc));
            HIPCHK(ctx->sbmg_flag.alloc(nS));
            HIPCHK(ctx->sbmg_cflag.alloc(nsc));
            HIPCHK(ctx->sbmg_work.alloc(5 * 4 * nsc));
            if (ctx->solid_coarse_exact) FSICHK(bcr_plan(ctx, nsc, scptr, sccol, nullptr));     // (leaves ctx->bcr null when the level does not suit it)
          } else {
            ctx->solid_mg = 0;
          }
        }
      } else {
        ctx->dd_mg = 0;
      }
    }
    HIPCHK(ctx->vvf_dinv32.alloc(4 * N2));
    HIPCHK(ctx->Avp.alloc(3 * padj_total));
    HIPCHK(ctx->Apv.alloc(rowptr_pv[V]));
    HIPCHK(ctx->App.alloc(rowptr_pp[V]));
    for (SubMat* M : {&ctx->Mdd, &ctx->Mvv}) {
      M->n = 3 * N2; M->nnz = 9 * nadj_total;
      M->rowptr = ctx->rowptr3.p; M->cols = ctx->cols3.p; M->diagpos = ctx->diagpos3.p;
      HIPCHK(M->vals.alloc(M->nnz));
    }
    {   // pattern of the explicit Schur complement: vertices that share a velocity node's element neighbourhood
      std::vector<int64_t> s_rowptr(V + 1, 0), s_diagpos(V, 0);
      std::vector<int32_t> s_cols, mark(V, -1), row;
      s_cols.reserve((size_t)V * 64);
      for (int64_t q = 0; q < V; ++q) {
        const int32_t r = prow_rank[q];
        row.clear();
        for (int64_t kb = ctx->h_nadj_ptr[r]; kb < ctx->h_nadj_ptr[r + 1]; ++kb) {
          const int32_t b = ctx->h_nadj[kb];
          for (int64_t t = ctx->h_padj_ptr[b]; t < ctx->h_padj_ptr[b + 1]; ++t) {
            const int32_t u = ctx->h_padj[t];
            if (mark[u] != (int32_t)q) { mark[u] = (int32_t)q; row.push_back(u); }
          }
        }
        std::sort(row.begin(), row.end());
        for (size_t t = 0; t < row.size(); ++t)
          if (row[t] == (int32_t)q) s_diagpos[q] = (int64_t)s_cols.size() + (int64_t)t;
        s_cols.insert(s_cols.end(), row.begin(), row.end());
        s_rowptr[q + 1] = (int64_t)s_cols.size();
      }
      FSICHK(upload(ctx, ctx->s_rowptr, s_rowptr));
      FSICHK(upload(ctx, ctx->s_diagpos, s_diagpos));
      FSICHK(upload(ctx, ctx->s_cols, s_cols));
      HIPCHK(ctx->s_vals.alloc(s_cols.size()));
      {   // tiles of the Schur pattern for k_sweep_schur_tiled: per 256 rows the distinct columns and 16-bit local indices
        // rows per tile (FsiTuning.schur_tile_rows; 0 = 64).  A sweep is a chain of dependent steps per workgroup (stage the tile's
        // distinct columns, 32 rows per pass, update) and the data is on die, so shorter chains in more workgroups win: round 4
        // scan, ms per preconditioner application with 64 / 128 / 256 rows: 1.10 / 1.13 / 1.28 at 140 k tets (24 k rows: 94 tiles
        // of 256 rows are fewer than the chip has CUs), 3.97 / 4.02 / 3.99 at 1.12 M tets (profiles/r04_tile_scan.txt)
        const int want = ctx->tune.schur_tile_rows;
        const int TR = (want == 32 || want == 64 || want == 128 || want == 256) ? want : 64;
        ctx->schur_tile = TR;
        const int64_t ntiles = (V + TR - 1) / TR;
        std::vector<int64_t> uptr(ntiles + 1, 0);
        std::vector<int32_t> ulist, tmpu;
        std::vector<uint16_t> ploc(s_cols.size());
        bool ok = true;
        int max_nu = 0;
        for (int64_t t = 0; t < ntiles && ok; ++t) {
          const int64_t q0 = t * TR, q1 = std::min<int64_t>(V, q0 + TR);
          const int64_t e0 = s_rowptr[q0], e1 = s_rowptr[q1];
          tmpu.assign(s_cols.begin() + e0, s_cols.begin() + e1);
          std::sort(tmpu.begin(), tmpu.end());
          tmpu.erase(std::unique(tmpu.begin(), tmpu.end()), tmpu.end());
          if (tmpu.size() > 7000) { ok = false; break; }          // 56 KB of LDS as doubles
          max_nu = std::max<int>(max_nu, (int)tmpu.size());
          for (int64_t e = e0; e < e1; ++e)
            ploc[e] = (uint16_t)(std::lower_bound(tmpu.begin(), tmpu.end(), s_cols[e]) - tmpu.begin());
          ulist.insert(ulist.end(), tmpu.begin(), tmpu.end());
          uptr[t + 1] = (int64_t)ulist.size();
        }
        ctx->schur_tiled = ok && V > 0;
        ctx->s_tile_max_nu = max_nu;
        if (ctx->schur_tiled) {
          FSICHK(upload(ctx, ctx->s_tile_uptr, uptr));
          FSICHK(upload(ctx, ctx->s_tile_ulist, ulist));
          FSICHK(upload(ctx, ctx->s_ploc, ploc));
        }
      }
    }
    HIPCHK(ctx->blk.alloc((size_t)20 * 3 * N2 + 16));
  }
  HIPCHK(hipMemsetAsync(ctx->A_pre.p, 0, ctx->nnz * sizeof(double), ctx->stream));
  HIPCHK(ctx->iflags.alloc(n + 16));
  // Krylov space: sized from free memory (the recycled directions are what 288 GB of HBM are used for)
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  ctx->kry_fp32 = ctx->kry_fp32_policy == 1;
  if (ctx->op32_policy && ctx->kry_fp32_policy != 0) {
    // layout of the FP32 copy (see k_spmv_node6p): node blocks padded to multiples of four entries, pressure rows behind
    std::vector<int64_t> rp(6 * (size_t)ctx->N2 + 2), p32((size_t)ctx->N2 + 1, 0);
    HIPCHK(hipMemcpy(rp.data(), ctx->rowptr.p, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < ctx->N2; ++r) {
      const int64_t L = rp[6 * r + 1] - rp[6 * r];
      p32[r + 1] = p32[r] + 6 * ((L + 3) & ~(int64_t)3);
    }
    ctx->a32_ptail = p32[ctx->N2];
    ctx->a32_tail_src = rp[6 * (size_t)ctx->N2];
    ctx->a32_tail_nnz = ctx->nnz - ctx->a32_tail_src;
    FSICHK(upload(ctx, ctx->a32_ptr, p32));
    HIPCHK(ctx->a32_cols.alloc((size_t)(ctx->a32_ptail / 6)));
    launch_pad_cols32(ctx->stream, ctx->N2, ctx->rowptr.p, ctx->cols.p, ctx->a32_ptr.p, ctx->a32_cols.p);
    HIPCHK(ctx->A32.alloc((size_t)(ctx->a32_ptail + ctx->a32_tail_nnz)));
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
  }
  ctx->debug_gcr = getenv("FSI_DEBUG_GCR") != nullptr;
  ctx->debug_prec_apply = (getenv("FSI_DEBUG_PRECOND") && atoi(getenv("FSI_DEBUG_PRECOND")) >= 2) ? 12 : 0;
  ctx->ldq = (n + 3) & ~(int64_t)3;
  ctx->ldz = (n + 1) & ~(int64_t)1;
  const double per_dir = (double)ctx->ldz * 8.0 + (double)ctx->ldq * (ctx->kry_fp32_policy == 1 ? 4.0 : 8.0);
  int64_t cap = (int64_t)((double)free_b * 0.5 / per_dir);
  // 600 kept directions (round 2: 400): a Jacobian's life of 20 steps makes ~380 early in a run and ~550 once the ramp is up
  // (4.6 Newton iterations per step); a full store rotates, and the 100-step run is 5 % faster without that (12.0 against
  // 11.4 Newton-it/s); the 20-step bench does not notice.  Half of the free HBM remains the upper limit.
  cap = std::max<int64_t>(8, std::min<int64_t>(cap, ctx->tune.krylov_capacity));
  ctx->kry_cap = cap;
  HIPCHK(ctx->KZ.alloc((size_t)cap * ctx->ldz));
  HIPCHK(ctx->KQ.alloc((size_t)cap * ctx->ldq * (ctx->kry_fp32_policy == 1 ? 4 : 8)));      // FP64-sized unless FP32 is forced
  HIPCHK(ctx->hcoef.alloc(cap + 4));
  if (ctx->kry_fp32_policy != 0) { HIPCHK(ctx->KQh.alloc((size_t)32 * ctx->ldq)); HIPCHK(ctx->hcoef_hot.alloc(40)); }
  ctx->hot_slots.assign(32, -1);
  HIPCHK(ctx->gcr_out.alloc(8));
  HIPCHK(ctx->gcr_y.alloc(cap));
  HIPCHK(ctx->gcr_cn.alloc((size_t)32 * cap));
  HIPCHK(ctx->gcr_slots.alloc(32));
  HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->gcr_host), (size_t)(cap + 64) * sizeof(double), hipHostMallocDefault));
  ctx->kry_born.assign(cap, -1);
  gcr_reset(ctx);
  HIPCHK(ctx->scratch.alloc(std::max<size_t>(24576, (size_t)(cap + 2) * 256 + 16)));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return FSI_OK;
}

int fsi_set_dirichlet(FsiCtx* ctx, int64_t n, const int64_t* dofs) {
  if (!ctx || n < 0 || (n > 0 && !dofs)) return FSI_ERR_INVALID;
  std::vector<int32_t> s(n);
  for (int64_t i = 0; i < n; ++i) {
    if (dofs[i] < 0 || dofs[i] >= ctx->ndof) { ctx->err = "fsi_set_dirichlet: dof out of range"; return FSI_ERR_INVALID; }
    s[i] = ctx->h_user2solver[dofs[i]];
  }
  ctx->nbc = n;
  ctx->h_bc = s;
  FSICHK(rebuild_matrix_bc(ctx));
  FSICHK(upload(ctx, ctx->bc_dofs, s));
  HIPCHK(ctx->bc_vals.alloc(n));
  if (n) HIPCHK(hipMemset(ctx->bc_vals.p, 0, n * sizeof(double)));
  return FSI_OK;
}

int fsi_set_dirichlet_values(FsiCtx* ctx, int64_t n, const double* values) {
  if (!ctx || n != ctx->nbc || (n > 0 && !values)) { if (ctx) ctx->err = "fsi_set_dirichlet_values: size mismatch"; return FSI_ERR_INVALID; }
  if (n) HIPCHK(hipMemcpy(ctx->bc_vals.p, values, n * sizeof(double), hipMemcpyHostToDevice));
  return FSI_OK;
}

int fsi_set_partition(FsiCtx* ctx, int64_t num_owned_cells, int64_t n_ghost, const int64_t* ghost_dofs,
                      int64_t n_identity, const int64_t* identity_dofs, int64_t n_send, const int64_t* send_dofs,
                      double* sendbuf_dev, double* recvbuf_dev, const FsiComm* comm) {
  if (!ctx) return FSI_ERR_INVALID;
  if (!comm || !comm->allreduce_sum || !comm->halo_exchange || num_owned_cells < 0 || num_owned_cells > ctx->C ||
      n_ghost < 0 || n_send < 0 || n_identity < 0 || n_identity > n_ghost || (n_ghost > 0 && (!ghost_dofs || !recvbuf_dev)) ||
      (n_identity > 0 && !identity_dofs) || (n_send > 0 && (!send_dofs || !sendbuf_dev))) {
    ctx->err = "fsi_set_partition: bad arguments";
    return FSI_ERR_INVALID;
  }
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<int32_t> g(n_ghost), idn(n_identity), sd(n_send);
  std::vector<uint8_t> is_ghost(ctx->ndof, 0);
  for (int64_t i = 0; i < n_ghost; ++i) {
    if (ghost_dofs[i] < 0 || ghost_dofs[i] >= ctx->ndof || is_ghost[ghost_dofs[i]]) { ctx->err = "fsi_set_partition: ghost dof out of range or repeated"; return FSI_ERR_INVALID; }
    is_ghost[ghost_dofs[i]] = 1;
    g[i] = ctx->h_user2solver[ghost_dofs[i]];
  }
  for (int64_t i = 0; i < n_identity; ++i) {
    if (identity_dofs[i] < 0 || identity_dofs[i] >= ctx->ndof || !is_ghost[identity_dofs[i]]) { ctx->err = "fsi_set_partition: identity dof is not a ghost dof"; return FSI_ERR_INVALID; }
    idn[i] = ctx->h_user2solver[identity_dofs[i]];
  }
  for (int64_t i = 0; i < n_send; ++i) {
    if (send_dofs[i] < 0 || send_dofs[i] >= ctx->ndof || is_ghost[send_dofs[i]]) { ctx->err = "fsi_set_partition: send dof out of range or not owned"; return FSI_ERR_INVALID; }
    sd[i] = ctx->h_user2solver[send_dofs[i]];
  }
  ctx->h_ident = idn;
  ctx->nghost = n_ghost;
  ctx->nident = n_identity;
  ctx->nsend = n_send;
  ctx->C_owned = num_owned_cells;
  FSICHK(upload(ctx, ctx->ghost_idx, g));
  FSICHK(upload(ctx, ctx->ident_idx, idn));
  FSICHK(upload(ctx, ctx->send_idx, sd));
  HIPCHK(ctx->ghost_zero.alloc(n_ghost));
  if (n_ghost) HIPCHK(hipMemset(ctx->ghost_zero.p, 0, n_ghost * sizeof(double)));
  FSICHK(rebuild_matrix_bc(ctx));
  ctx->sendbuf = sendbuf_dev;
  ctx->recvbuf = recvbuf_dev;
  ctx->comm = *comm;
  ctx->part = true;
  ctx->have_jacobian = false;
  gcr_reset(ctx);
  // whether the preconditioner sees the residual on an overlap is one decision for the whole job (a collective halo
  // update hangs if some ranks skip it): any rank with complete ghost rows switches it on for all
  double overlap = ctx->nident < ctx->nghost ? 1.0 : 0.0;
  if (ctx->comm.allreduce_sum(ctx->comm.user, &overlap, 1) != 0) { ctx->err = "allreduce callback failed"; return FSI_ERR_DEVICE; }
  ctx->ras = overlap > 0.0;
  return FSI_OK;
}

int fsi_rccl_unique_id(void* id128) {
  if (!id128) return FSI_ERR_INVALID;
  std::string err;
  return rccl_unique_id(id128, &err);
}

int fsi_set_rccl(FsiCtx* ctx, const void* id128, int32_t rank, int32_t world, const int64_t* send_counts, const int64_t* recv_counts) {
  if (!ctx) return FSI_ERR_INVALID;
  if (!id128) {                // back to the FsiComm callbacks of fsi_set_partition (the communicator, if any, is destroyed)
    HIPCHK(hipSetDevice(ctx->device));
    rccl_destroy(ctx);
    return FSI_OK;
  }
  if (world < 1 || rank < 0 || rank >= world || !send_counts || !recv_counts) { ctx->err = "fsi_set_rccl: bad arguments"; return FSI_ERR_INVALID; }
  return rccl_init(ctx, id128, rank, world, send_counts, recv_counts);
}

int fsi_set_pressure_facets(FsiCtx* ctx, int64_t nf, const int32_t* facet_nodes, const int32_t* plus_cell) {
  if (!ctx || nf < 0 || (nf > 0 && (!facet_nodes || !plus_cell))) return FSI_ERR_INVALID;
  std::vector<int32_t> dofs;
  std::vector<double> coef;
  const double* X = ctx->h_coords.data();
  for (int64_t f = 0; f < nf; ++f) {
    const int32_t* fn = facet_nodes + 6 * f;
    const int32_t cell = plus_cell[f];
    if (cell < 0 || cell >= ctx->C) { ctx->err = "fsi_set_pressure_facets: bad cell"; return FSI_ERR_INVALID; }
    for (int a = 0; a < 6; ++a)
      if (fn[a] < 0 || fn[a] >= ctx->N2 || (a < 3 && fn[a] >= ctx->V)) { ctx->err = "fsi_set_pressure_facets: bad node"; return FSI_ERR_INVALID; }
    const double *a = X + 3 * fn[0], *b = X + 3 * fn[1], *c = X + 3 * fn[2];
    double e1[3], e2[3], nv[3], fc[3], cc[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) { e1[i] = b[i] - a[i]; e2[i] = c[i] - a[i]; fc[i] = (a[i] + b[i] + c[i]) / 3.0; }
    nv[0] = e1[1] * e2[2] - e1[2] * e2[1];
    nv[1] = e1[2] * e2[0] - e1[0] * e2[2];
    nv[2] = e1[0] * e2[1] - e1[1] * e2[0];       // |nv| = 2 area
    for (int v = 0; v < 4; ++v)
      for (int i = 0; i < 3; ++i) cc[i] += 0.25 * X[3 * ctx->h_tet_nodes[10 * (int64_t)cell + v] + i];
    const double sgn = (nv[0] * (fc[0] - cc[0]) + nv[1] * (fc[1] - cc[1]) + nv[2] * (fc[2] - cc[2])) < 0 ? -1.0 : 1.0;
    // int N_a ds = 0 for the vertex functions, area/3 for the edge functions (P2 triangle)
    for (int e = 3; e < 6; ++e)
      for (int i = 0; i < 3; ++i) {
        dofs.push_back(6 * ctx->h_node2rank[fn[e]] + 3 + i);
        coef.push_back(sgn * 0.5 * nv[i] / 3.0);
      }
  }
  // one entry per dof (a node's facets summed here, in facet order): the device adds each target once, so the load vector
  // does not depend on the order in which atomics arrive
  {
    std::vector<int64_t> order(dofs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int64_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return dofs[a] < dofs[b]; });
    std::vector<int32_t> udofs;
    std::vector<double> ucoef;
    for (int64_t i : order) {
      if (!udofs.empty() && udofs.back() == dofs[i]) ucoef.back() += coef[i];
      else { udofs.push_back(dofs[i]); ucoef.push_back(coef[i]); }
    }
    dofs.swap(udofs);
    coef.swap(ucoef);
  }
  ctx->npf = (int64_t)dofs.size();
  FSICHK(upload(ctx, ctx->pf_dofs, dofs));
  FSICHK(upload(ctx, ctx->pf_coef, coef));
  return FSI_OK;
}

int fsi_set_interface_pressure(FsiCtx* ctx, double P) {
  if (!ctx) return FSI_ERR_INVALID;
  ctx->P = P;
  return FSI_OK;
}

int fsi_set_robin_facets(FsiCtx* ctx, int64_t nf, const int32_t* facet_nodes, const double* k_s, const double* c_s) {
  if (!ctx || nf < 0 || (nf > 0 && (!facet_nodes || !k_s || !c_s))) return FSI_ERR_INVALID;
  // reference P2 triangle mass matrix / area: (1/180) [[6,-1,-1,-4,0,0],[-1,6,-1,0,-4,0],[-1,-1,6,0,0,-4],
  //                                                  [-4,0,0,32,16,16],[0,-4,0,16,32,16],[0,0,-4,16,16,32]]
  static const double M[6][6] = {{6, -1, -1, -4, 0, 0},  {-1, 6, -1, 0, -4, 0},  {-1, -1, 6, 0, 0, -4},
                                 {-4, 0, 0, 32, 16, 16}, {0, -4, 0, 16, 32, 16}, {0, 0, -4, 16, 16, 32}};
  std::vector<int32_t> row, col;
  std::vector<double> val;
  std::vector<int64_t> pos;
  const double* X = ctx->h_coords.data();
  std::vector<int64_t> h_rowptr(ctx->ndof + 1);
  HIPCHK(hipMemcpy(h_rowptr.data(), ctx->rowptr.p, (ctx->ndof + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  for (int64_t f = 0; f < nf; ++f) {
    const int32_t* fn = facet_nodes + 6 * f;
    for (int a = 0; a < 6; ++a)
      if (fn[a] < 0 || fn[a] >= ctx->N2 || (a < 3 && fn[a] >= ctx->V)) { ctx->err = "fsi_set_robin_facets: bad node"; return FSI_ERR_INVALID; }
    const double *a = X + 3 * fn[0], *b = X + 3 * fn[1], *c = X + 3 * fn[2];
    double e1[3], e2[3];
    for (int i = 0; i < 3; ++i) { e1[i] = b[i] - a[i]; e2[i] = c[i] - a[i]; }
    const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
    const double area = 0.5 * std::sqrt(nx * nx + ny * ny + nz * nz);
    for (int p = 0; p < 6; ++p)
      for (int q = 0; q < 6; ++q) {
        const double m = area * M[p][q] / 180.0;
        if (m == 0.0) continue;
        const int32_t rp = ctx->h_node2rank[fn[p]], rq = ctx->h_node2rank[fn[q]];
        const int32_t* lo = ctx->h_nadj.data() + ctx->h_nadj_ptr[rp];
        const int32_t* hi = ctx->h_nadj.data() + ctx->h_nadj_ptr[rp + 1];
        const int64_t k = std::lower_bound(lo, hi, rq) - lo;
        for (int i = 0; i < 3; ++i) {
          const int32_t r = 6 * rp + 3 + i;
          row.push_back(r); col.push_back(6 * rq + i);     val.push_back(k_s[f] * m); pos.push_back(h_rowptr[r] + 6 * k + i);
          row.push_back(r); col.push_back(6 * rq + 3 + i); val.push_back(c_s[f] * m); pos.push_back(h_rowptr[r] + 6 * k + 3 + i);
        }
      }
  }
  // sorted by (row, column) with duplicates merged: one thread per row adds its entries in that order (residual), and every
  // matrix position is added once (A_pre) - no dependence on the order of atomics
  {
    std::vector<int64_t> order(row.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int64_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return row[a] != row[b] ? row[a] < row[b] : col[a] < col[b]; });
    std::vector<int32_t> r2, c2, urow, ptr;
    std::vector<double> v2;
    std::vector<int64_t> p2;
    for (int64_t i : order) {
      if (!r2.empty() && r2.back() == row[i] && c2.back() == col[i]) { v2.back() += val[i]; continue; }
      if (r2.empty() || r2.back() != row[i]) { urow.push_back(row[i]); ptr.push_back((int32_t)r2.size()); }
      r2.push_back(row[i]); c2.push_back(col[i]); v2.push_back(val[i]); p2.push_back(pos[i]);
    }
    ptr.push_back((int32_t)r2.size());
    row.swap(r2); col.swap(c2); val.swap(v2); pos.swap(p2);
    ctx->nrobin_rows = (int64_t)urow.size();
    FSICHK(upload(ctx, ctx->rb_urow, urow));
    FSICHK(upload(ctx, ctx->rb_ptr, ptr));
  }
  ctx->nrobin = (int64_t)row.size();
  FSICHK(upload(ctx, ctx->rb_row, row));
  FSICHK(upload(ctx, ctx->rb_col, col));
  FSICHK(upload(ctx, ctx->rb_val, val));
  FSICHK(upload(ctx, ctx->rb_pos, pos));
  return FSI_OK;
}
static double* state_ptr(FsiCtx* ctx, int which) {
  switch (which) {
    case 0: return ctx->U.p;
    case 1: return ctx->U1.p;
    case 2: return ctx->b.p;
    case 3: return ctx->du.p;
    default: return nullptr;
  }
}

int fsi_get_state(FsiCtx* ctx, int which, double* out) {
  if (!ctx || !out || !state_ptr(ctx, which)) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  launch_gather(ctx->stream, ctx->tmp7.p, state_ptr(ctx, which), ctx->user2solver.p, ctx->ndof);   // tmp7[user] = x[solver]
  HIPCHK(hipMemcpyAsync(out, ctx->tmp7.p, ctx->ndof * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return FSI_OK;
}

int fsi_get_values(FsiCtx* ctx, int which, int64_t n, const int64_t* dofs, double* out) {
  if (!ctx || !state_ptr(ctx, which) || n < 0 || (n > 0 && (!dofs || !out))) return FSI_ERR_INVALID;
  if (n == 0) return FSI_OK;
  if (n > ctx->ndof) { ctx->err = "fsi_get_values: more dofs than the problem has"; return FSI_ERR_INVALID; }
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<int32_t> idx((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (dofs[i] < 0 || dofs[i] >= ctx->ndof) { ctx->err = "fsi_get_values: dof out of range"; return FSI_ERR_INVALID; }
    idx[i] = ctx->h_user2solver[dofs[i]];
  }
  if (ctx->gv_idx.n < (size_t)n) HIPCHK(ctx->gv_idx.alloc((size_t)n));
  HIPCHK(hipMemcpyAsync(ctx->gv_idx.p, idx.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  launch_gather(ctx->stream, ctx->tmp7.p, state_ptr(ctx, which), ctx->gv_idx.p, n);
  HIPCHK(hipMemcpyAsync(out, ctx->tmp7.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return FSI_OK;
}

int fsi_set_state(FsiCtx* ctx, int which, const double* in) {
  if (!ctx || !in || !state_ptr(ctx, which)) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(ctx->tmp7.p, in, ctx->ndof * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  launch_scatter(ctx->stream, state_ptr(ctx, which), ctx->tmp7.p, ctx->user2solver.p, ctx->ndof);  // x[solver] = in[user]
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return FSI_OK;
}

int fsi_get_matrix(FsiCtx* ctx, int64_t* rowptr, int64_t* cols, double* vals) {
  if (!ctx || !rowptr || !cols || !vals) return FSI_ERR_INVALID;
  if (!ctx->have_jacobian) { ctx->err = "fsi_get_matrix: no Jacobian assembled"; return FSI_ERR_INVALID; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->ndof, nnz = ctx->nnz;
  std::vector<int64_t> rp(n + 1);
  std::vector<int32_t> cs(nnz), s2u(n);
  std::vector<double> vs(nnz), sc(n);
  HIPCHK(hipMemcpy(rp.data(), ctx->rowptr.p, (n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cs.data(), ctx->cols.p, nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(vs.data(), ctx->A.p, nnz * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(sc.data(), ctx->rowscale.p, n * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(s2u.data(), ctx->solver2user.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  rowptr[0] = 0;
  for (int64_t u = 0; u < n; ++u) {
    const int64_t s = ctx->h_user2solver[u];
    rowptr[u + 1] = rowptr[u] + (rp[s + 1] - rp[s]);
  }
  std::vector<std::pair<int64_t, double>> rowbuf;
  for (int64_t u = 0; u < n; ++u) {
    const int64_t s = ctx->h_user2solver[u];
    rowbuf.clear();
    for (int64_t t = rp[s]; t < rp[s + 1]; ++t) rowbuf.emplace_back((int64_t)s2u[cs[t]], vs[t] / sc[s]);
    std::sort(rowbuf.begin(), rowbuf.end());
    int64_t o = rowptr[u];
    for (auto& e : rowbuf) { cols[o] = e.first; vals[o] = e.second; ++o; }
  }
  return FSI_OK;
}

int fsi_spmv(FsiCtx* ctx, const double* x, double* y) {
  if (!ctx || !x || !y) return FSI_ERR_INVALID;
  if (!ctx->have_jacobian) { ctx->err = "fsi_spmv: no Jacobian assembled"; return FSI_ERR_INVALID; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->ndof;
  HIPCHK(hipMemcpyAsync(ctx->tmp7.p, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  launch_scatter(ctx->stream, ctx->tmp1.p, ctx->tmp7.p, ctx->user2solver.p, n);
  FSICHK(spmv(ctx, ctx->tmp1.p, ctx->tmp2.p));                 // the kernel of the outer Krylov method
  // undo the row equilibration: y = D^-1 (D A) x
  launch_gather(ctx->stream, ctx->tmp7.p, ctx->tmp2.p, ctx->user2solver.p, n);
  std::vector<double> ys(n), sc(n);
  HIPCHK(hipMemcpyAsync(ys.data(), ctx->tmp7.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  launch_gather(ctx->stream, ctx->tmp3.p, ctx->rowscale.p, ctx->user2solver.p, n);
  HIPCHK(hipMemcpyAsync(sc.data(), ctx->tmp3.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int64_t i = 0; i < n; ++i) y[i] = ys[i] / sc[i];
  return FSI_OK;
}

int fsi_apply_preconditioner(FsiCtx* ctx, const double* r, double* z) {
  if (!ctx || !r || !z) return FSI_ERR_INVALID;
  if (!ctx->have_jacobian) { ctx->err = "fsi_apply_preconditioner: no Jacobian assembled"; return FSI_ERR_INVALID; }
  if (ctx->part) { ctx->err = "fsi_apply_preconditioner: single contexts only (a partitioned one applies its rank-local part inside fsi_solve)"; return FSI_ERR_INVALID; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->ndof;
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(ctx->tmp7.p, r, n * sizeof(double), hipMemcpyHostToDevice, st));
  launch_scatter(st, ctx->tmp1.p, ctx->tmp7.p, ctx->user2solver.p, n);
  launch_mul(st, ctx->tmp1.p, ctx->tmp1.p, ctx->rowscale.p, n);      // the solver works on D A x = D b
  FSICHK(precondition(ctx, ctx->tmp1.p, ctx->tmp2.p));
  launch_gather(st, ctx->tmp7.p, ctx->tmp2.p, ctx->user2solver.p, n);
  HIPCHK(hipMemcpyAsync(z, ctx->tmp7.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return FSI_OK;
}

int fsi_probe(FsiCtx* ctx, int64_t n, const int32_t* cells, const double* bary, double* out) {
  if (!ctx || n < 0 || (n > 0 && (!cells || !bary || !out))) return FSI_ERR_INVALID;
  if (n == 0) return FSI_OK;
  for (int64_t i = 0; i < n; ++i)
    if (cells[i] < 0 || cells[i] >= ctx->C) { ctx->err = "fsi_probe: cell out of range (locate the points first)"; return FSI_ERR_INVALID; }
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<int32_t> dc;
  DevBuf<double> db, dout;
  HIPCHK(dc.alloc(n)); HIPCHK(db.alloc(4 * n)); HIPCHK(dout.alloc(7 * n));
  HIPCHK(hipMemcpyAsync(dc.p, cells, n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(db.p, bary, 4 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  launch_probe(ctx->stream, n, elem_arrays(ctx), dc.p, db.p, ctx->U.p, dout.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, dout.p, 7 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dc.release(); db.release(); dout.release();
  return FSI_OK;
}

int fsi_flow_stats(FsiCtx* ctx, double* out) {
  if (!ctx || !out) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->cellvals.p) HIPCHK(ctx->cellvals.alloc(2 * ctx->C + 8 + 4 * STAT_PARTS));
  double* res = ctx->cellvals.p + 2 * ctx->C;
  // partitioned: the cells this rank owns (they come first); the host combines the ranks with the owned-cell counts
  const int64_t nc = ctx->part ? ctx->C_owned : ctx->C;
  if (nc <= 0) { out[0] = 0.0; out[1] = 1e300; out[2] = -1e300; out[3] = 1e300; return FSI_OK; }
  launch_cell_stats(ctx->stream, nc, elem_arrays(ctx), ctx->U.p, ctx->cellvals.p, res);
  HIPCHK(hipGetLastError());
  double h[4];
  HIPCHK(hipMemcpyAsync(h, res, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  out[0] = h[0] / (double)nc; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
  return FSI_OK;
}

int fsi_calibration_streams(FsiCtx* ctx, int64_t bytes) {
  if (!ctx || bytes <= 0) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t have = (int64_t)ctx->KZ.n * 8;
  if (bytes > have) bytes = have;
  bytes &= ~(int64_t)255;
  launch_calibration(ctx->stream, ctx->KZ.p, bytes, ctx->gcr_out.p);      // the direction store is scratch between solves
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  gcr_reset(ctx);
  return FSI_OK;
}

int fsi_stress_strain(FsiCtx* ctx, int64_t n, const int32_t* cells, double* out) {
  if (!ctx || n < 0 || (n > 0 && (!cells || !out))) return FSI_ERR_INVALID;
  if (n == 0) return FSI_OK;
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<int32_t> kinds((size_t)ctx->C);
  HIPCHK(hipMemcpy(kinds.data(), ctx->cell_kind.p, (size_t)ctx->C * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) {
    if (cells[i] < 0 || cells[i] >= ctx->C) { ctx->err = "fsi_stress_strain: cell out of range"; return FSI_ERR_INVALID; }
    if (kinds[cells[i]] != 1) { ctx->err = "fsi_stress_strain: cell is not a solid cell"; return FSI_ERR_INVALID; }
  }
  DevBuf<int32_t> dc;
  DevBuf<double> dout;
  HIPCHK(dc.alloc((size_t)n));
  HIPCHK(dout.alloc((size_t)n * 80));
  HIPCHK(hipMemcpyAsync(dc.p, cells, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  launch_stress_strain(ctx->stream, n, elem_arrays(ctx), elem_params(ctx), ctx->U.p, dc.p, dout.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * 80 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dc.release();
  dout.release();
  return FSI_OK;
}

int fsi_wall_shear_stress(FsiCtx* ctx, int64_t nf, const int32_t* facet_cells, const int32_t* facet_local, double mu,
                          double* out) {
  if (!ctx || nf < 0 || (nf > 0 && (!facet_cells || !facet_local || !out)) || !(mu > 0.0)) return FSI_ERR_INVALID;
  if (nf == 0) return FSI_OK;
  HIPCHK(hipSetDevice(ctx->device));
  // one projection per boundary cell: a cell with several exterior facets couples them through its shared vertices
  std::vector<int32_t> ucell, mask, slot((size_t)nf);
  {
    std::vector<std::pair<int32_t, int64_t>> order((size_t)nf);
    for (int64_t f = 0; f < nf; ++f) {
      if (facet_cells[f] < 0 || facet_cells[f] >= ctx->C || facet_local[f] < 0 || facet_local[f] > 3) {
        ctx->err = "fsi_wall_shear_stress: facet cell / local index out of range";
        return FSI_ERR_INVALID;
      }
      order[f] = {facet_cells[f], f};
    }
    std::sort(order.begin(), order.end());
    for (int64_t k = 0; k < nf; ++k) {
      if (k == 0 || order[k].first != order[k - 1].first) { ucell.push_back(order[k].first); mask.push_back(0); }
      mask.back() |= 1 << facet_local[order[k].second];
      slot[order[k].second] = (int32_t)ucell.size() - 1;
    }
  }
  const int64_t nc = (int64_t)ucell.size();
  DevBuf<int32_t> dc, dm;
  DevBuf<double> dout;
  HIPCHK(dc.alloc((size_t)nc));
  HIPCHK(dm.alloc((size_t)nc));
  HIPCHK(dout.alloc((size_t)nc * 12));
  HIPCHK(hipMemcpyAsync(dc.p, ucell.data(), (size_t)nc * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(dm.p, mask.data(), (size_t)nc * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  launch_wss(ctx->stream, nc, elem_arrays(ctx), ctx->U.p, dc.p, dm.p, mu, dout.p);
  HIPCHK(hipGetLastError());
  std::vector<double> h((size_t)nc * 12);
  HIPCHK(hipMemcpyAsync(h.data(), dout.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  static const int VERTS[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
  for (int64_t f = 0; f < nf; ++f)
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < 3; ++i) out[(f * 3 + k) * 3 + i] = h[((size_t)slot[f] * 4 + VERTS[facet_local[f]][k]) * 3 + i];
  dc.release(); dm.release(); dout.release();
  return FSI_OK;
}

int64_t fsi_xcd_order(int64_t n, int64_t* unit_out) {
  if (n < 0) return -1;
  const int64_t span = fsi::xcd_span(n);
  if (unit_out)
    for (int64_t L = 0; L < span; ++L) unit_out[L] = fsi::xcd_unit(L, n);
  return span;
}

int fsi_get_solver_events(const FsiCtx* ctx, int64_t out[8]) {
  if (!ctx || !out) return FSI_ERR_INVALID;
  out[0] = ctx->ev_base[0] + ctx->newton_retries;
  out[1] = ctx->ev_base[1] + ctx->kry_fp32_failures_total;
  out[2] = ctx->ev_base[2] + ctx->gcr_restarts;
  out[3] = ctx->newton_adaptive_solves; out[4] = ctx->utol_tightened; out[5] = ctx->bcr_solves;
  out[6] = out[7] = 0;
  return FSI_OK;
}

int fsi_get_timers(FsiCtx* ctx, FsiTimers* out, int reset) {
  if (!ctx || !out) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  for (PhaseTimer* t : {&ctx->t_res, &ctx->t_jac, &ctx->t_fac, &ctx->t_spmv, &ctx->t_prec, &ctx->t_ortho, &ctx->t_flush, &ctx->t_kry}) resolve_timer(t, t->issued);
  *out = FsiTimers{ctx->t_res.ms,  ctx->t_res.calls,  ctx->t_jac.ms,   ctx->t_jac.calls,   ctx->t_fac.ms, ctx->t_fac.calls,
                   ctx->t_spmv.ms, ctx->t_spmv.calls, ctx->t_prec.ms,  ctx->t_prec.calls,  ctx->t_ortho.ms,
                   ctx->t_ortho.calls, ctx->t_kry.ms, ctx->t_kry.calls, ctx->kry_iters,
                   ctx->inner_its[0], ctx->inner_its[1], ctx->inner_its[2], ctx->inner_calls,
                   ctx->t_ss.ms, ctx->t_ss.calls, ctx->solid_fp32 ? 9 * ctx->sb_nblocks : (int64_t)ctx->ss_vals.n, 3 * ctx->nS,
                   ctx->t_db.ms, ctx->t_db.calls, (int64_t)ctx->dd_db.n / 3, ctx->N2, ctx->t_sc.ms, ctx->t_sc.calls,
                   (int64_t)(ctx->dd_is_scalar && ctx->sweeps_fp32) + (ctx->tiled ? 2 : 0), (int64_t)ctx->tile_ulist.n,
                   ctx->ortho_q_cols, ctx->ortho_q_launches, ctx->ortho_z_cols, ctx->ortho_z_launches,
                   (int64_t)(ctx->kry_fp32 ? 4 : 8), ctx->ldq, ctx->ldz, ctx->kry_hw, ctx->kry_cap,
                   (int64_t)ctx->s_cols.n, ctx->V, ctx->t_flush.ms, ctx->t_flush.calls, ctx->t_sch.ms, ctx->t_sch.calls,
                   (int64_t)((ctx->schur_fp32 && ctx->s_vals32.p) ? ((ctx->schur_tiled && ctx->sweeps_fp16 && ctx->s_rec.p) ? 0 : 4) : 8),
                   (int64_t)ctx->h_nadj.size(), (int64_t)ctx->h_padj.size(),
                   ctx->op32_products,
                   (int64_t)((ctx->tiled && ctx->fused_sweeps ? 1 : 0) | (ctx->tiled && ctx->fused_sweeps && ctx->sweeps_fp16 ? 2 : 0) |
                             (ctx->solid_fp32 ? 4 : 0) | (ctx->solid_fp32 && ctx->solid_block_jacobi && ctx->solid_fused ? 8 : 0) |
                             (ctx->sbmg_ready ? 16 : 0) | (ctx->mg_ready ? 32 : 0) | (ctx->drows_ok ? 64 : 0)),
                   ctx->part_allreduces, (int64_t)ctx->ncellcol, ctx->gcr_arnoldi_steps, ctx->gcr_restarts, ctx->newton_retries,
                   (int64_t)ctx->kry_fp32_failures_total, ctx->verdicts_skipped, ctx->gcr_reorth_forced, ctx->dd_cache_hits,
                   ctx->newton_late_solves};
  if (reset) {
    for (PhaseTimer* t : {&ctx->t_res, &ctx->t_jac, &ctx->t_fac, &ctx->t_spmv, &ctx->t_prec, &ctx->t_ortho, &ctx->t_flush, &ctx->t_sch, &ctx->t_kry, &ctx->t_ss, &ctx->t_db, &ctx->t_sc}) {
      t->ms = 0.0;
      t->calls = 0;
    }
    ctx->kry_iters = 0;
    ctx->op32_products = 0;
    ctx->inner_its[0] = ctx->inner_its[1] = ctx->inner_its[2] = 0;
    ctx->inner_calls = 0;
    ctx->ortho_q_cols = ctx->ortho_q_launches = ctx->ortho_z_cols = ctx->ortho_z_launches = 0;
    ctx->part_allreduces = 0;
    ctx->ev_base[0] += ctx->newton_retries; ctx->ev_base[1] += ctx->kry_fp32_failures_total; ctx->ev_base[2] += ctx->gcr_restarts;   // run totals survive
    ctx->gcr_arnoldi_steps = ctx->gcr_restarts = ctx->newton_retries = ctx->kry_fp32_failures_total = 0;
    ctx->verdicts_skipped = ctx->gcr_reorth_forced = 0;
    ctx->dd_cache_hits = ctx->newton_late_solves = 0;
    ctx->sample_budget = 16;      // the sweep kernels of the next 16 preconditioner applications are sampled with events
  }
  return FSI_OK;
}

}  // extern "C"
