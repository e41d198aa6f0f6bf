// Declarations shared by the HIP translation units of libvaspfsi.so.
#pragma once
#include "fsi_context.hpp"

namespace fsi {
#if defined(__HIPCC__)
// XCD-aware order of a launch's logical workgroups.  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2; a tile's
// staged neighbour entries are mostly those of the tiles beside it along the (Morton) node order.  Logical workgroup L (XCD L & 7,
// slot L >> 3 on that XCD) takes unit (L & 7) * ceil(n / 8) + (L >> 3): XCD k works through the k-th eighth of the units, as
// k_spmv_node6 does, so that what neighbouring units share is fetched into ONE L2 instead of eight.  Measured on one box against the
// launch order and against runs of 16 / 64 consecutive units dealt to the XCDs in turn (NOTEBOOK.md section 9): L2 -> fabric bytes of
// k_residual 2.29 -> 1.65 x the algorithmic ones, Schur sweep 1.87 -> 1.55, solid fine sweep 1.12 -> 1.00; step time unchanged at 1.12 M tets (the sweeps are paced
// by dependent loads, not bytes), -4 % at 140 k; the three mappings equal.
// xcd_span(n) logical workgroups cover n units (a multiple of 8); xcd_unit returns -1 for a logical workgroup without a unit.
__host__ __device__ inline int64_t xcd_span(int64_t n) { return (n + 7) / 8 * 8; }
__host__ __device__ inline int64_t xcd_unit(int64_t L, int64_t n) {
  const int64_t chunk = (n + 7) >> 3, s = L >> 3, t = (L & 7) * chunk + s;
  return s < chunk && t < n ? t : -1;
}
static inline unsigned xcd_grid(int64_t n) { return (unsigned)xcd_span(n); }
// Sum over an aligned group of 4 / 8 / 16 lanes by DPP adds (quad permutes, then half-row and row mirrors): four VALU
// instructions per value instead of four ds_bpermute round trips through the LDS crossbar, which is what __shfl_xor
// compiles to and what paced the 16-lanes-per-row sweep kernels.  Every lane of the group ends up with the sum.
template <int CTRL>
__device__ inline float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ inline double dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
template <int LANES>
__device__ inline float group_sum(float v) {
  v += dpp_f<0xB1>(v);                          // quad_perm [1,0,3,2]
  if (LANES >= 4) v += dpp_f<0x4E>(v);          // quad_perm [2,3,0,1]
  if (LANES >= 8) v += dpp_f<0x141>(v);         // row_half_mirror: the other quad of the half row (quads are uniform by now)
  if (LANES >= 16) v += dpp_f<0x140>(v);        // row_mirror: the other half row
  return v;
}
template <int LANES>
__device__ inline double group_sum(double v) {
  v += dpp_d<0xB1>(v);
  if (LANES >= 4) v += dpp_d<0x4E>(v);
  if (LANES >= 8) v += dpp_d<0x141>(v);
  if (LANES >= 16) v += dpp_d<0x140>(v);
  return v;
}

// Sum over the 64 lanes of a wave: the four row sums by DPP, then four lane reads.  The result is wave-uniform.
__device__ inline double wave_sum_dpp(double v) {
  v = group_sum<16>(v);
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  double t = 0.0;
#pragma unroll
  for (int l = 0; l < 64; l += 16) {
    const int rl = __builtin_amdgcn_readlane(lo, l), rh = __builtin_amdgcn_readlane(hi, l);
    t += __builtin_bit_cast(double, ((long long)rh << 32) | (long long)(unsigned int)rl);
  }
  return t;
}
#endif

struct ElemArrays {
  const double* geom;          // [C][10]
  const int32_t* cell_dofs;    // [C][64]
  const int32_t* cell_kind;    // [C]
  const int32_t* cell_region;  // [C]
  const int32_t* cell_rank;    // [C][10]
  const int32_t* cell_prow;    // [C][4] solver-layout dofs of the four pressure unknowns of a cell
  const uint16_t* enbr;        // [C][10][10]
  const uint16_t* epnbr;       // [C][10][4]
};

// Assembly colouring: cells of one colour share no node, so one launch over a colour adds at most once to any entry of the
// global vector / matrix and the colours' launches are ordered by the stream: assembly is bitwise reproducible.
struct CellColours {
  int ncolours = 0;              // 0: one launch over all cells, unordered atomics
  const int32_t* cells = nullptr;   // device: cell ids sorted by colour (ascending inside a colour)
  const int64_t* ptr = nullptr;     // host: [ncolours + 1]
};

// Residual assembly as a gather: element vectors in Re[C][64], summed per dof over the incidences of its node (see k_residual)
struct ResidualGather {
  double* Re = nullptr;             // nullptr: scatter-add with atomics instead
  int64_t N2 = 0, V = 0;
  const int64_t* inc_ptr = nullptr;    // [N2 + 1] by node rank
  const int32_t* inc = nullptr;        // 16 * cell + local node, ascending
  const int64_t* pinc_ptr = nullptr;   // [V + 1] by pressure row
  const int32_t* pinc = nullptr;       // 16 * cell + local vertex, ascending
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
                     const double* U1, double* F, const ResidualGather& rg);
constexpr int STAT_PARTS = 256;      // stage-1 workgroups of the cell-statistics reduction: cellvals needs 2 C + 8 + 4 * STAT_PARTS doubles
void launch_cell_stats(hipStream_t st, int64_t C, const ElemArrays& ea, const double* X, double* cellvals, double* out);
void launch_probe(hipStream_t st, int64_t n, const ElemArrays& ea, const int32_t* cells, const double* bary, const double* X,
                  double* out);
void launch_l2norm(hipStream_t st, int64_t C, const ElemArrays& ea, const double* X, double* part, double* out);
void launch_jacobian(hipStream_t st, int part, int64_t C, const ElemArrays& ea, const ElemParams& ep, const double* U,
                     const double* U1, const int64_t* rowptr, const int64_t* nadj_ptr, double* vals, const CellColours& cc, int jac_waves = 2, int jac_mfma = 0);

void launch_sweep_csr_f64(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const double* vals,
                          const int64_t* diagpos, double c1, double c2, const double* din, double* dout, double* x, double* r);
void launch_f32_ripple4(hipStream_t st, int64_t nnodes, float* x);                 // pseudo-random float4 per node, pad lane 0
void launch_f32_sumsq(hipStream_t st, int64_t n, const float* x, double* out);      // out[0] = sum x^2, fixed summation order

// fsi_rccl.hip — collectives issued by the library on the solver stream (RCCL resolved with dlopen)
int rccl_unique_id(void* out128, std::string* err);
int rccl_init(FsiCtx* ctx, const void* id128, int rank, int world, const int64_t* send_counts, const int64_t* recv_counts);
void rccl_destroy(FsiCtx* ctx);
int rccl_allreduce_dev(FsiCtx* ctx, double* dptr, int64_t n);   // in place, device memory, on the solver stream
int rccl_allreduce_host(FsiCtx* ctx, double* v, int n);         // host values through a staging buffer (one wait)
int rccl_halo(FsiCtx* ctx);                                     // sendbuf -> peers' recvbuf, grouped send / recv

// fsi_post.hip — solid stress / strain and wall shear stress (cell-local DG1 projections)
hipError_t upload_post_tables(const double* qw, const double* dN, const double* L);
void launch_stress_strain(hipStream_t st, int64_t ncell, const ElemArrays& ea, const ElemParams& ep, const double* U,
                          const int32_t* cells, double* out);
void launch_wss(hipStream_t st, int64_t ncell, const ElemArrays& ea, const double* U, const int32_t* cells, const int32_t* fmask,
                double mu, double* out);

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
void launch_div(hipStream_t st, double* z, const double* x, const double* y, int64_t n);     // z = x ./ y
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
void launch_robin_residual(hipStream_t st, int64_t nrows, const int32_t* urow, const int32_t* ptr, const int32_t* col,
                           const double* val, double th0, double th1, const double* U, const double* U1, double* F);
void launch_add_at(hipStream_t st, double* vals, const int64_t* pos, const double* v, double a, int64_t n);
enum SpmvTag : int { SPMV_MONOLITHIC = 0, SPMV_SOLID_BLOCK = 1, SPMV_FIELD_BLOCK = 2 };
// the node graph the pressure rows of the monolithic matrix were laid out from (k_expand_cols): with it the products read one
// neighbour rank per six entries (k_spmv_prow); all null: the generic CSR kernel
struct PRowGraph {
  const int32_t* vrank = nullptr;
  const int64_t* nadj_ptr = nullptr;
  const int32_t* nadj = nullptr;
};
void launch_pad_cols32(hipStream_t st, int64_t N2, const int64_t* rowptr, const int32_t* cols, const int64_t* p32, int32_t* cols32);
void launch_pad_vals32(hipStream_t st, int64_t N2, int64_t V, const int64_t* rowptr, const double* A, const int64_t* p32,
                       int64_t ptail, int64_t nnz_tail, int64_t tail_src, float* A32, bool v_rows_only = false);
// d rows of the node blocks in pair form [dd_0 dd_1 dd_2 dv_0 dv_1 dv_2] per node pair (fsi_solver.hip); flag bit 0: a d row holds
// something else and the products must keep the full rows
void launch_drows_extract(hipStream_t st, int64_t N2, const int64_t* rowptr, const double* A, const int64_t* nadj_ptr, double* ad64,
                          float* ad32, int32_t* flag);
void launch_spmv_node6p(hipStream_t st, int64_t N2, int64_t V, const int64_t* p32, const int32_t* cols32, const float* vals,
                        const int64_t* rowptr, const int32_t* cols, int64_t tail_shift, const PRowGraph& g, const double* x, double* y,
                        const float* ad32 = nullptr);
void launch_round_to_f32(hipStream_t st, int64_t n, const double* a, float* b);
void launch_spmv(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const double* vals,
                 const double* x, double* y, int tag = SPMV_FIELD_BLOCK);
// reductions: out[0] = x.y (deterministic two-stage); scratch must hold >= 4096 doubles
void launch_dot(hipStream_t st, const double* x, const double* y, int64_t n, double* scratch, double* out);
void launch_hashed_sum(hipStream_t st, const double* x, int64_t first, int64_t stride, int64_t n, double* scratch, double* out);
// known-byte read / write streams (4, 8, 16, 32 bytes per lane) over buf[0, bytes): PMC counter calibration
void launch_calibration(hipStream_t st, void* buf, int64_t bytes, double* out);
// fsi_gcr.hip — orthogonalisation against the kept directions of the recycled GCR (Q in FP32 or FP64, see there)
// out[k] = Q_k . w (k < m), out[m] = w . w, out[m+1] = w . r (r may be null); scratch >= (m + 2) * 64 doubles
void launch_gcr_dots(hipStream_t st, bool fp32, const void* Q, int64_t ldq, int64_t n, int m, const double* w,
                     const double* r, double* scratch, double* out);
// w -= sum_k h[k] Q_k; out2[0] = |w'|^2, out2[1] = w' . r
void launch_gcr_axpy(hipStream_t st, bool fp32, const void* Q, int64_t ldq, int64_t n, int m, const double* h, double* w,
                     const double* r, double* scratch, double* out2);
// Q_slot = qd = w * inv_wn, Z_slot = z, r -= alpha w * inv_wn; out1[0] = |r'|^2
void launch_gcr_update(hipStream_t st, bool fp32, void* Q, int64_t ldq, double* Z, int64_t ldz, int slot, int64_t n,
                       const double* w, const double* z, double inv_wn, double alpha, double* r, double* qd,
                       double* scratch, double* out1);
// x += sum_j y[j] Z_j; Z_slots[k] = sum_j cn[k * m + j] Z_j for k < knew (<= 32); cn has gcr_flush_width(knew) columns
int gcr_flush_width(int knew);
void launch_gcr_flush(hipStream_t st, double* Z, int64_t ldz, int64_t n, int m, const double* y, const double* cn,
                      const int32_t* slots, int knew, double* x);
// incomplete factorisation and triangular solves, colour by colour (fsi_solver.hip)
void launch_ilu0_levels(hipStream_t st, const std::vector<Level>& levels, const int64_t* rowptr, const int32_t* cols,
                        const int64_t* diagpos, double* LU, int32_t* counters);
void launch_sptrsv_levels(hipStream_t st, const std::vector<Level>& levels, const int64_t* rowptr, const int32_t* cols,
                          const int64_t* diagpos, const double* LU, const double* rhs, double* tmp, double* x);


// fsi_block.hip — field blocks of the Jacobian and the pieces of the block preconditioner
void launch_block_structure(hipStream_t st, int64_t N2, int64_t V, const int64_t* nadj_ptr, const int32_t* nadj,
                            const int64_t* padj_ptr, const int32_t* padj, const int32_t* vrank, int64_t* rowptr3,
                            int32_t* cols3, int64_t* diagpos3, int64_t* rowptr_vp, int32_t* cols_vp,
                            const int64_t* rowptr_pv, int32_t* cols_pv);
void launch_extract_blocks(hipStream_t st, int64_t N2, int64_t V, double ktheta, const int64_t* rowptr, const double* A,
                           const int64_t* nadj_ptr, const int32_t* nadj, const int64_t* padj_ptr, const int32_t* vrank,
                           const int32_t* node_solid, const int64_t* rowptr3, const int64_t* rowptr_vp,
                           const int64_t* rowptr_pv, const int64_t* rowptr_pp, double* Add, double* Adv, double* Avv,
                           double* Avp, double* Apv, double* App);
void launch_schur_full(hipStream_t st, int64_t V, const int64_t* s_rowptr, const int32_t* s_cols, const int32_t* vrank,
                       const int64_t* nadj_ptr, const int32_t* nadj, const int64_t* padj_ptr, const int32_t* padj,
                       const int64_t* rowptr_pv, const double* Apv, const int64_t* rowptr_pp, const double* App,
                       const int64_t* rowptr_vp, const double* Avp, const int64_t* diagpos3, const double* Avv, double* S,
                       int32_t* flags);
void launch_split(hipStream_t st, int64_t N2, int64_t V, const double* r, double* rd, double* rv, double* rp);
void launch_merge(hipStream_t st, int64_t N2, int64_t V, const double* zd, const double* zv, const double* zp, double* z);
void launch_vel_correct32(hipStream_t st, int64_t N2, const int64_t* padj_ptr, const int32_t* padj, const float* avp, const double* dp,
                          const double* dinv, const double* vs, double* dv);
void launch_pres_rhs32(hipStream_t st, int64_t V, const int32_t* vrank, const int64_t* nadj_ptr, const int32_t* nadj,
                       const int64_t* rowptr_pv, const float* apv, const double* w, const double* c, double* y);
void launch_vel_correct(hipStream_t st, int64_t n3, const int64_t* rowptr, const int32_t* cols, const double* vals,
                        const double* dp, const int64_t* diagpos3, const double* Avv, const double* vs, double* dv,
                        const double* dinv = nullptr);
void launch_diag_inverse(hipStream_t st, int64_t n, const int64_t* diagpos, const double* A, double* dinv);
void launch_pres_rows(hipStream_t st, int64_t V, const int64_t* rowptr_pp, const int32_t* cols_pp, const double* App,
                      const double* x, double alpha, const int64_t* rowptr_pv, const int32_t* cols_pv, const double* Apv,
                      const double* w, double beta, const double* c, double gamma, double* y);
void launch_cheb_init(hipStream_t st, int64_t n, const double* mask, const double* rhs, const int64_t* diagpos,
                      const double* A, double inv_theta, double* x, double* r, double* d);
void launch_cheb_step(hipStream_t st, int64_t n, const double* mask, const double* t, const int64_t* diagpos,
                      const double* A, double c1, double c2, double* x, double* r, double* d);
void launch_extract_db(hipStream_t st, int64_t N2, int64_t npairs, const int64_t* nadj_ptr, const int64_t* rowptr3,
                       const double* vals, double* db, int32_t* flags, int check);
void launch_pad_init_f32(hipStream_t st, int64_t nn, const double* a, const float* scale4, const float* dinv4, float inv_theta,
                         float* x, float* r, float* d);
void launch_merge_f32d(hipStream_t st, int64_t N2, int64_t V, const float* xd4, const double* zv, const double* zp, double* z);
void launch_mask_outside(hipStream_t st, int64_t N2, const uint8_t* rowmask, const int32_t* node_set, int32_t* flags);
void launch_db_rows_sub(hipStream_t st, int64_t nl, const int32_t* list, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                        const uint8_t* rowmask, const double* x, double* y);
void launch_spmv_db(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                    const double* x, double* y, const uint8_t* rowmask = nullptr);
void launch_db_rowmask(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const double* db, uint8_t* rowmask);
void launch_spmv_tiled_f32(hipStream_t st, int nv, int tn, int64_t N2, int max_nu, const int64_t* nadj_ptr, const float* vals,
                           const uint16_t* ploc, const int64_t* tile_uptr, const int32_t* ulist, const uint8_t* rowflag,
                           const float* x, float* y);
int tile_limit();
void launch_extract_chat(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                         float* chat, uint8_t* rowflag, int32_t* flags);
void launch_sweep_tiled_f32(hipStream_t st, int nv, int tn, int64_t N2, int max_nu, const int64_t* nadj_ptr, const float* vals,
                            const uint16_t* ploc, const int64_t* tile_uptr, const int32_t* ulist, const uint8_t* rowflag,
                            const float* dinv, float c1, float c2, const float* din, float* dout, float* x, float* r);
void launch_pack_h1(hipStream_t st, int64_t n, const float* v, const uint16_t* loc, uint32_t* rec);
void launch_pack_h3(hipStream_t st, int64_t n, const float* v, const uint16_t* loc, void* rec);
void launch_pack_sb(hipStream_t st, int64_t nb, const float* v, const int32_t* col, void* rec);
void launch_sweep_tiled_h(hipStream_t st, int nv, int tn, int64_t N2, int max_nu, const int64_t* nadj_ptr, const void* rec,
                          const int64_t* tile_uptr, const int32_t* ulist, const uint8_t* rowflag, const float* dinv, float c1,
                          float c2, const float* din, float* dout, float* x, float* r);
void launch_sweep_sb_h(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const void* rec, const float* binv12, float c1, float c2,
                       const float* din, float* dout, float* x, float* r);
void launch_sweep_sc_f32(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const float* chat,
                         const uint8_t* rowflag, float c1, float c2, const float* din, float* dout, float* x, float* r);
void launch_spmv_sc_f32(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const float* chat,
                        const uint8_t* rowflag, const float* x, float* y);
void launch_spmv_db_f32(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const float* db,
                        const float* x, float* y);
void launch_to_f32(hipStream_t st, int64_t n, const double* a, float* b);
void launch_pad_to_f32(hipStream_t st, int64_t nn, const double* a, const float* scale4, float* b);
void launch_unpad_from_f32(hipStream_t st, int64_t nn, const float* a, double* b);
void launch_dinv_f32(hipStream_t st, int64_t n, const double* mask, const int64_t* diagpos, const double* A, float* dinv);
void launch_sb_binv(hipStream_t st, int64_t nS, const int32_t* snode, const int64_t* diagpos3, const double* Avv,
                    float* binv12, double* binv9);
void launch_block_scale_d(hipStream_t st, int64_t nS, const double* binv9, double* y);
void launch_cheb_init_b3(hipStream_t st, int64_t nS, const float* rhs, const float* binv12, float inv_theta, float* x, float* r, float* d);
void launch_mg_d0(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                  const double* rowscale, const uint8_t* rowflag, float* d0, int32_t* flags);
void launch_mg_rap(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                   const int64_t* nadj_ptr, const int32_t* nadj, const double* db, const double* rowscale,
                   const uint8_t* rowflag, const int32_t* par, const float* pw, const int64_t* cptr, const int32_t* ccol,
                   double* Ac, int32_t* flags);
void launch_mg_coarse_finish(hipStream_t st, int64_t nc, const int64_t* cptr, const int32_t* ccol, const double* Ac,
                             const int32_t* cfine, const uint8_t* rowflag, float* cc, uint8_t* cflag, float* dcinv4,
                             int32_t* rowmax_bits);
void launch_mg_restrict(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                        const float* d0, const float* r4, const float* dcinv4, float* rc4, float inv_theta = 0.f, float* cx = nullptr, float* cr = nullptr,
                        float* cd = nullptr);
void launch_mg_prolong(hipStream_t st, int64_t N2, const int32_t* par, const float* pw, const float* d0, const float* xc4,
                       float* e4);
void launch_residual_rows(hipStream_t st, int64_t nrows, const int32_t* rows, const int64_t* ptr, const int32_t* col,
                          const int64_t* src, const double* vals, const double* x, const double* b, double* y);
void launch_spmv_node6(hipStream_t st, int64_t N2, int64_t V, const int64_t* rowptr, const int32_t* cols, const double* vals,
                       const PRowGraph& g, const double* x, double* y, const double* ad64 = nullptr);
void launch_sweep_csr_mixed(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const float* vals,
                            const int64_t* diagpos, const double* dvals, double c1, double c2, const double* din, double* dout,
                            double* x, double* r);
void launch_sweep_schur_tiled(hipStream_t st, int tile_rows, int64_t n, int max_nu, const int64_t* rowptr, const uint32_t* rec,
                              const int64_t* tile_uptr, const int32_t* ulist, const double* dinv, double c1, double c2,
                              const double* din, double* dout, double* x, double* r);
void launch_sbmg_flags(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const int32_t* sb_col, const float* vals, uint8_t* flag);
void launch_sbmg_rap(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                     const int64_t* sb_ptr, const int32_t* sb_col, const float* vals, const int32_t* snode,
                     const double* rowscale, const uint8_t* flag, const int32_t* par, const float* pw, const int64_t* cptr,
                     const int32_t* ccol, float* cvals, int32_t* flags);
void launch_sbmg_coarse_finish(hipStream_t st, int64_t nc, const int64_t* cptr, const int32_t* ccol, float* cvals,
                               const int32_t* cfine, const uint8_t* flag, float* cbinv12, uint8_t* cflag, int32_t* rowmax_bits);
void launch_sbmg_restrict(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                          const int32_t* snode, const double* rowscale, const uint8_t* flag, const uint8_t* cflag,
                          const float* r4, float* rc4, const int32_t* bpos = nullptr, double* bd = nullptr);
void launch_sbmg_prolong(hipStream_t st, int64_t nS, const int32_t* par, const float* pw, const uint8_t* flag, const float* xc4,
                         float* e4, const int32_t* bpos = nullptr, const double* xd = nullptr);
void launch_sweep_sb_b3(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const int32_t* sb_col, const float* vals,
                        const float* binv12, float c1, float c2, const float* din, float* dout, float* x, float* r,
                        int level = 0);
void launch_cheb_step_b3(hipStream_t st, int64_t nS, const float* t, const float* binv12, float c1, float c2, float* x, float* r, float* d);
void launch_sb_gather(hipStream_t st, int64_t nb, const int32_t* sb_row, const int64_t* sb_src, const int32_t* sb_stride,
                      const double* Avv, float* vals);
void launch_sb_dinv(hipStream_t st, int64_t nS, const int32_t* snode, const int64_t* diagpos3, const double* Avv, float* dinv);
void launch_spmv_sb(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const int32_t* sb_col, const float* vals,
                    const float* x, float* y);
void launch_cheb_init_f32(hipStream_t st, int64_t n, const float* rhs, const float* dinv, float inv_theta, float* x, float* r, float* d);
void launch_cheb_step_f32(hipStream_t st, int64_t n, const float* t, const float* dinv, float c1, float c2, float* x, float* r, float* d);
void launch_gather3_f32(hipStream_t st, int64_t nS, const int32_t* snode, const double* full, float* comp);
void launch_solid_cycle_init(hipStream_t st, int64_t nS, const int32_t* snode, const double* full, const float* binv12, float scale,
                             float* x, float* r, float* d, float* d2);
void launch_scatter3_f32(hipStream_t st, int64_t nS, const int32_t* snode, const float* comp, double* full);
void launch_gather_vals(hipStream_t st, int64_t n, const int64_t* pos, const double* src, double* dst);
void launch_gather3(hipStream_t st, int64_t nS, const int32_t* snode, const double* full, double* comp);
void launch_scatter3(hipStream_t st, int64_t nS, const int32_t* snode, const double* comp, double* full);
void launch_mask_ripple(hipStream_t st, int64_t n, const double* mask, double* x);
void launch_mask_scale(hipStream_t st, int64_t n, const double* mask, const int64_t* diagpos, const double* A, double* y);
void launch_residual_csr(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const double* vals,
                         const double* x, const double* b, double* y);

}  // namespace fsi
