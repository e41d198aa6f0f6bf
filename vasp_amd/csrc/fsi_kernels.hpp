// Declarations shared by the HIP translation units of libvaspfsi.so.
#pragma once
#include "fsi_context.hpp"

namespace fsi {

struct ElemArrays {
  const double* geom;          // [C][10]
  const int32_t* cell_dofs;    // [C][64]
  const int32_t* cell_kind;    // [C]
  const int32_t* cell_region;  // [C]
  const int32_t* cell_rank;    // [C][10]
  const uint16_t* enbr;        // [C][10][10]
  const uint16_t* epnbr;       // [C][10][4]
};

struct ElemParams {
  Scheme sc;
  FluidProps fluid[MAX_REGIONS];
  SolidProps solid[MAX_REGIONS];
};

// fsi_assembly.hip
hipError_t upload_tables();
void launch_geometry(hipStream_t st, int64_t C, const double* coords, const int32_t* tet_vertices, double* geom);
void launch_residual(hipStream_t st, int64_t C, const ElemArrays& ea, const ElemParams& ep, const double* U,
                     const double* U1, double* F);
void launch_jacobian(hipStream_t st, int part, int64_t C, const ElemArrays& ea, const ElemParams& ep, const double* U,
                     const double* U1, const int64_t* rowptr, const int64_t* nadj_ptr, double* vals);

// fsi_solver.hip — sparse / dense vector kernels
void launch_expand_cols(hipStream_t st, int64_t N2, int64_t V, const int64_t* nadj_ptr, const int32_t* nadj,
                        const int64_t* padj_ptr, const int32_t* padj, const int32_t* prow_rank, const int64_t* rowptr,
                        int32_t* cols, int64_t* diagpos);
void launch_fill(hipStream_t st, double* x, int64_t n, double v);
void launch_copy(hipStream_t st, double* dst, const double* src, int64_t n);
void launch_axpy(hipStream_t st, double* y, double a, const double* x, int64_t n);           // y += a x
void launch_axpby(hipStream_t st, double* z, double a, const double* x, double b, const double* y, int64_t n);
void launch_scale(hipStream_t st, double* y, double a, int64_t n);
void launch_mul(hipStream_t st, double* z, const double* x, const double* y, int64_t n);     // z = x .* y
void launch_gather(hipStream_t st, double* dst, const double* src, const int32_t* idx, int64_t n);   // dst[i]=src[idx[i]]
void launch_scatter(hipStream_t st, double* dst, const double* src, const int32_t* idx, int64_t n);  // dst[idx[i]]=src[i]
void launch_add_indexed(hipStream_t st, double* y, const int32_t* idx, const double* coef, double a, int64_t n);
// rhs finalisation: b = -F; b[bc] = g - U[bc]
void launch_negate(hipStream_t st, double* b, const double* F, int64_t n);
void launch_bc_rhs(hipStream_t st, double* b, const double* U, const int32_t* bc, const double* g, int64_t nbc);
void launch_bc_set(hipStream_t st, double* U, const int32_t* bc, const double* g, int64_t nbc);
// matrix finishing: A = Jn + Apre; ident_zeros; bc rows; row equilibration
void launch_matrix_finish(hipStream_t st, int64_t n, const int64_t* rowptr, const int64_t* diagpos, double* A,
                          const double* Apre, const int32_t* bc, int64_t nbc, double* rowscale, int32_t* bcmask);
void launch_robin_residual(hipStream_t st, int64_t n, const int32_t* row, const int32_t* col, const double* val,
                           double th0, double th1, const double* U, const double* U1, double* F);
void launch_add_at(hipStream_t st, double* vals, const int64_t* pos, const double* v, double a, int64_t n);
void launch_spmv(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const double* vals,
                 const double* x, double* y);
// reductions: out[0] = x.y (deterministic two-stage); scratch must hold >= 4096 doubles
void launch_dot(hipStream_t st, const double* x, const double* y, int64_t n, double* scratch, double* out);
// h[i] = Q_i . w for i < m  (Q stored as m contiguous vectors of length n); then w -= sum_i h[i] Q_i
void launch_multi_dot(hipStream_t st, const double* Q, int64_t n, int m, const double* w, double* scratch, double* h);
void launch_multi_axpy(hipStream_t st, const double* Q, int64_t n, int m, const double* h, double sign, double* w);
// incomplete factorisation and triangular solves, colour by colour (fsi_solver.hip)
void launch_ilu0_levels(hipStream_t st, const std::vector<Level>& levels, const int64_t* rowptr, const int32_t* cols,
                        const int64_t* diagpos, double* LU, int32_t* counters);
void launch_sptrsv_levels(hipStream_t st, const std::vector<Level>& levels, const int64_t* rowptr, const int32_t* cols,
                          const int64_t* diagpos, const double* LU, const double* rhs, double* tmp, double* x);

}  // namespace fsi
