// FsiTuning (include/vaspfsi.h): defaults and the ONE place where the library reads product options from the environment.
//
// Rounds 1-3 grew ~80 getenv() switches across five files; what survived measurement is a field of FsiTuning, what did not
// (dense third level, Krylov-space compression, compact node rows, column-free product, all-FP32 Schur sweeps, inner ILU /
// BiCGStab solves of the field blocks) left the library in round 4.  FSI_<NAME> overrides the field <name>; the debugging aids
// FSI_DEBUG* are read where they are used and are not options.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/vaspfsi.h"

extern "C" {

}  // extern "C"

namespace {
// struct_size is in / out (include/vaspfsi.h): a caller built against a shorter, older FsiTuning announces its sizeof there
// and only that many bytes of its buffer are written; 0 (a zeroed struct) or anything outside (0, sizeof] means "this header".
size_t caller_bytes(const FsiTuning* t) {
  const int32_t sz = t->struct_size;
  return (sz >= (int32_t)sizeof(int32_t) && sz <= (int32_t)sizeof(FsiTuning)) ? (size_t)sz : sizeof(FsiTuning);
}
void hand_out(const FsiTuning& full, FsiTuning* t, size_t n) {
  std::memcpy(t, &full, n);
  t->struct_size = (int32_t)n;
}
void defaults(FsiTuning* t);
void from_env(FsiTuning* t);
}  // namespace

extern "C" {

void fsi_tuning_defaults(FsiTuning* t) {
  if (!t) return;
  const size_t n = caller_bytes(t);
  FsiTuning full;
  defaults(&full);
  hand_out(full, t, n);
}

void fsi_tuning_from_env(FsiTuning* t) {
  if (!t) return;
  const size_t n = caller_bytes(t);
  FsiTuning full;
  from_env(&full);
  hand_out(full, t, n);
}

// fsi_get_tuning (fsi_capi.hip) hands the context's struct out through this, under the same rule
void fsi_tuning_copy_out(const FsiTuning* full, FsiTuning* out) {
  if (!full || !out) return;
  hand_out(*full, out, caller_bytes(out));
}

}  // extern "C"

namespace {

void defaults(FsiTuning* t) {
  std::memset(t, 0, sizeof *t);
  t->struct_size = (int32_t)sizeof(FsiTuning);
  t->krylov_fp32 = 2; t->operator_fp32 = 1; t->schur_fp32 = 1; t->sweeps_fp32 = 1; t->sweeps_fp16 = 1; t->solid_fp32 = 1; t->pv_fp32 = 1;
  t->krylov_capacity = 600; t->krylov_fp32_floor = 1e-10;
  t->assembly_atomic = 0; t->node_order = 0; t->tiles = 1; t->jacobian_waves = 2; t->jacobian_mfma = 0;
  t->newton_forcing = 1e-2; t->newton_forcing_late = 3e-3; t->newton_late_factor = 10.0;
  t->f32_cycle_floor = 1e-6; t->f32_verdict_skip_rtol = 3e-4; t->orth_floor32 = 3e-7; t->orth_floor64 = 1e-9;
  t->gcr_escape = 1e-3; t->gcr_reorth = 0.0;
  t->prec_streams = 1; t->cheb4 = 1; t->coarse_power = 1; t->solid_mg = 1; t->dd_mg = 1; t->mg_keep = 1;
  t->solid_block_jacobi = 1; t->solid_fused = 1; t->fused_sweeps = 1; t->scalar_dd = 1;
  t->its_solid = 300; t->its_fluid = 4; t->its_schur = 30; t->its_disp = 60;
  t->kappa_solid = 1e4; t->kappa_fluid = 5.0; t->kappa_schur = 100.0; t->kappa_disp = 1000.0;
  t->sbmg_pre = 16; t->sbmg_post = 16; t->sbmg_cits = 90; t->sbmg_alpha = 200.0; t->sbmg_ckappa = 4000.0;
  t->mg_pre = 3; t->mg_post = 0; t->mg_cits = 16; t->mg_alpha = 20.0; t->mg_ckappa = 250.0;
  t->solid_coarse_exact = 1; t->bcr_shift = 2e-4; t->newton_adaptive = 0.3; t->compact_drows = 1;
}

void from_env(FsiTuning* t) {
  defaults(t);
  auto I = [](const char* name, int32_t* v) { if (const char* e = getenv(name)) *v = (int32_t)atoi(e); };
  auto D = [](const char* name, double* v) { if (const char* e = getenv(name)) *v = atof(e); };
  I("FSI_KRYLOV_FP32", &t->krylov_fp32); I("FSI_OPERATOR_FP32", &t->operator_fp32); I("FSI_SCHUR_FP32", &t->schur_fp32);
  I("FSI_SWEEPS_FP32", &t->sweeps_fp32); I("FSI_SWEEPS_FP16", &t->sweeps_fp16); I("FSI_SOLID_FP32", &t->solid_fp32);
  I("FSI_PV_FP32", &t->pv_fp32); I("FSI_KRYLOV_CAP", &t->krylov_capacity); D("FSI_KRYLOV_FP32_FLOOR", &t->krylov_fp32_floor);
  if (const char* e = getenv("FSI_ASSEMBLY")) t->assembly_atomic = std::string(e) == "atomic";
  if (const char* e = getenv("FSI_ORDER")) t->node_order = (e[0] == 'c' || e[0] == 'C') ? 2 : (e[0] == 'm' || e[0] == 'M') ? 1 : 0;
  if (getenv("FSI_NO_TILES")) t->tiles = 0;
  I("FSI_SCHUR_TILE", &t->schur_tile_rows); I("FSI_TILE_NODES", &t->tile_nodes);
  if (getenv("FSI_NO_SCALAR_DD")) t->scalar_dd = 0;
  I("FSI_JAC_WAVES", &t->jacobian_waves); I("FSI_JAC_MFMA", &t->jacobian_mfma);
  D("FSI_NEWTON_FORCING", &t->newton_forcing); D("FSI_NEWTON_FORCING_LATE", &t->newton_forcing_late);
  D("FSI_NEWTON_LATE_FACTOR", &t->newton_late_factor); D("FSI_F32_CYCLE_FLOOR", &t->f32_cycle_floor);
  D("FSI_F32_VERDICT_SKIP", &t->f32_verdict_skip_rtol); D("FSI_ORTH_FLOOR32", &t->orth_floor32); D("FSI_ORTH_FLOOR64", &t->orth_floor64);
  D("FSI_GCR_ESCAPE", &t->gcr_escape); D("FSI_GCR_REORTH", &t->gcr_reorth);
  I("FSI_PREC_STREAMS", &t->prec_streams); I("FSI_EXPERIMENT", &t->experiment); I("FSI_CHEB4", &t->cheb4); I("FSI_COARSE_POWER", &t->coarse_power);
  I("FSI_SOLID_MG", &t->solid_mg); I("FSI_DD_MG", &t->dd_mg); I("FSI_MG_KEEP", &t->mg_keep); I("FSI_SOLID_BJ", &t->solid_block_jacobi);
  I("FSI_SOLID_FUSED", &t->solid_fused); I("FSI_FUSED_SWEEPS", &t->fused_sweeps);
  I("FSI_CHEB_S", &t->its_solid); I("FSI_CHEB_F", &t->its_fluid); I("FSI_CHEB_P", &t->its_schur); I("FSI_CHEB_D", &t->its_disp);
  D("FSI_KAPPA_S", &t->kappa_solid); D("FSI_KAPPA_F", &t->kappa_fluid); D("FSI_KAPPA_P", &t->kappa_schur); D("FSI_KAPPA_D", &t->kappa_disp);
  I("FSI_SBMG_PRE", &t->sbmg_pre); I("FSI_SBMG_POST", &t->sbmg_post); I("FSI_SBMG_CITS", &t->sbmg_cits);
  D("FSI_SBMG_ALPHA", &t->sbmg_alpha); D("FSI_SBMG_CKAPPA", &t->sbmg_ckappa);
  I("FSI_MG_PRE", &t->mg_pre); I("FSI_MG_POST", &t->mg_post); I("FSI_MG_CITS", &t->mg_cits);
  D("FSI_MG_ALPHA", &t->mg_alpha); D("FSI_MG_CKAPPA", &t->mg_ckappa);
  I("FSI_SOLID_COARSE_EXACT", &t->solid_coarse_exact); D("FSI_BCR_SHIFT", &t->bcr_shift); D("FSI_NEWTON_ADAPTIVE", &t->newton_adaptive);
  I("FSI_COMPACT_DROWS", &t->compact_drows);
}

}  // namespace
