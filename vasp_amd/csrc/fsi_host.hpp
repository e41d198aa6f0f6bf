// Host side of libvaspfsi.so: what the translation units behind the C-ABI share.
//   fsi_capi.hip     the ABI itself: context set-up (fsi_create), boundary data, partition, state access, timers
//   fsi_newton.hip   fsi_assemble_residual / _jacobian, fsi_solve, fsi_newton_solve (turtleFSI's newtonsolver policy)
//   fsi_krylov.hip   recycled GCR (solve_gcr), BiCGStab, the monolithic product
//   fsi_precond.hip  the field-split block preconditioner: one application (two streams), refresh at a new Jacobian
// The kernels are in fsi_assembly / fsi_solver / fsi_block / fsi_gcr / fsi_post .hip (declared in fsi_kernels.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "fsi_kernels.hpp"

#define HIPCHK(call)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
      return FSI_ERR_DEVICE;                                                                       \
    }                                                                                              \
  } while (0)
#define FSICHK(call)                 \
  do {                               \
    int r_ = (call);                 \
    if (r_ != FSI_OK) return r_;     \
  } while (0)

namespace fsi {
namespace host {

// HIP-event bracket on the solver stream.  Nothing here waits for the device: the pair goes into the timer's ring and is
// read back when the ring wraps (32 brackets later, long finished) or by resolve_timer() from fsi_get_timers.
inline void resolve_timer(PhaseTimer* t, int64_t upto) {
  for (; t->resolved < upto; ++t->resolved) {
    const int k = (int)(t->resolved % PhaseTimer::RING);
    float ms = 0.f;
    if (hipEventSynchronize(t->e1[k]) == hipSuccess && hipEventElapsedTime(&ms, t->e0[k], t->e1[k]) == hipSuccess) t->ms += ms;
  }
}
struct Phase {
  FsiCtx* c;
  PhaseTimer* t;
  int k;
  Phase(FsiCtx* ctx, PhaseTimer* tm) : c(ctx), t(tm) {
    if (t->issued - t->resolved >= PhaseTimer::RING) resolve_timer(t, t->issued - PhaseTimer::RING + 1);
    k = (int)(t->issued % PhaseTimer::RING);
    if (!t->e0[k]) { (void)hipEventCreate(&t->e0[k]); (void)hipEventCreate(&t->e1[k]); }
    (void)hipEventRecord(t->e0[k], c->stream);
  }
  ~Phase() {
    (void)hipEventRecord(t->e1[k], c->stream);
    t->issued += 1;
    t->calls += 1;
  }
};

template <class T>
int upload(FsiCtx* ctx, DevBuf<T>& buf, const std::vector<T>& h) {
  HIPCHK(buf.alloc(h.size()));
  if (!h.empty()) HIPCHK(hipMemcpy(buf.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return FSI_OK;
}

// fsi_capi.hip
ElemArrays elem_arrays(FsiCtx* c);
ResidualGather residual_gather(const FsiCtx* c);
CellColours cell_colours(const FsiCtx* c);
ElemParams elem_params(FsiCtx* c);
int host_scalar(FsiCtx* ctx, const double* dptr, double* out);
int dot_n(FsiCtx* ctx, const double* x, const double* y, int64_t n, double* out);
int dot(FsiCtx* ctx, const double* x, const double* y, double* out);
int norm2(FsiCtx* ctx, const double* x, double* out);
int allreduce(FsiCtx* ctx, double* v, int n);          // element partition: sum over the ranks (no-op in a single context)
int agree(FsiCtx* ctx, int rc);                        // ... a rank-local status made job-wide before the next collective
int gdot(FsiCtx* ctx, const double* x, const double* y, double* out);
int gnorm2(FsiCtx* ctx, const double* x, double* out);
int halo_update(FsiCtx* ctx, double* x);               // owner -> ghost refresh
void zero_ghost(FsiCtx* ctx, double* x);
int rebuild_matrix_bc(FsiCtx* ctx);
// fsi_precond.hip
int precondition(FsiCtx* ctx, const double* r, double* z);
int precondition_block(FsiCtx* ctx, const double* r, double* z);
int refresh_preconditioner(FsiCtx* ctx);
// fsi_bcr.hip: exact solve of the solid cycle's coarse level (block cyclic reduction)
struct BcrPlanStats {
  int64_t blocks = 0, levels = 0, bytes32 = 0, bytes64 = 0, setup_flops = 0;
  int max_block = 0, launches = 0, usable = 0, plan_only = 0;
  int32_t* pos_out = nullptr;      // plan_only: [nc] position of every node in BFS-level order
  int32_t* level_out = nullptr;    // plan_only: [nc] BFS level of every node
};
int bcr_plan(FsiCtx* ctx, int64_t nc, const std::vector<int64_t>& cptr, const std::vector<int32_t>& ccol, BcrPlanStats* stats);
int bcr_refresh(FsiCtx* ctx);
int bcr_solve(FsiCtx* ctx, const float* rc4, float* xc4, hipStream_t st);       // rc4 / xc4 == nullptr: the caller filled bcr_rhs / reads bcr_sol
const int32_t* bcr_pos(const FsiCtx* ctx);      // coarse node -> position in the solve's own (breadth-first) order
double* bcr_rhs(FsiCtx* ctx);                   // [3 nc] right-hand side / solution in that order (restriction writes, prolongation reads)
const double* bcr_sol(const FsiCtx* ctx);
bool bcr_ready(const FsiCtx* ctx);
void bcr_free(FsiCtx* ctx);
// fsi_krylov.hip
int spmv(FsiCtx* ctx, const double* x, double* y, bool working = false);
void gcr_reset(FsiCtx* ctx);
int solve_gcr(FsiCtx* ctx, const double* rhs, double* x, double rtol, int max_it, int* iters, double* relres);
int solve_bicgstab(FsiCtx* ctx, const double* rhs, double* x, double rtol, int max_it, int* iters, double* relres);

}  // namespace host
}  // namespace fsi
