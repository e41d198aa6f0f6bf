// Field-split pieces of the linear solve (gfx950): structure and values of the sub-matrices of the monolithic Jacobian
// and the pressure Schur complement used by the block preconditioner (fsi_capi.hip: precondition_block).
//
// The reference hands the monolithic matrix to MUMPS (`up_sol.solve`).  Here the Krylov method runs on
// the monolithic matrix and is preconditioned by an approximate block factorisation in the order (v, p) -> d:
//   * in the solid the d-equation  delta rho/k (d - d1) = delta rho (theta v + ...)  is a mass-matrix identity, so
//     dd = k theta dv there; substituting it turns A_vd, A_pd columns of solid nodes into velocity columns:
//       Avv~ = A_vv + k theta A_vd[:, solid],   Apv~ = A_pv + k theta A_pd[:, solid]
//   * the (v, p) saddle point is split SIMPLE-style with the explicit Schur complement S = A_pp - Apv~ D^-1 A_vp,
//     D = diag(Avv~), assembled on its full (two-ring) vertex pattern;
//   * d follows from A_dd dd = r_d - A_dv dv (solid: mass matrix; fluid: the mesh-lifting Laplacian).
// All of it is HBM-bound gather/scatter work; bytes are those of the matrices touched (stated at the launchers).
#include "fsi_kernels.hpp"

namespace fsi {

__device__ inline double wsum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- structure of the 3x3-blocked node matrices (A_dd, Avv~, A_dv share it) -------------------------------------------
// row 3r+i: for every neighbour rank s of r (ascending) the columns 3s, 3s+1, 3s+2.
__global__ void k_b3_structure(int64_t N2, const int64_t* __restrict__ nadj_ptr, const int32_t* __restrict__ nadj,
                               int64_t* __restrict__ rowptr3, int32_t* __restrict__ cols3, int64_t* __restrict__ diagpos3) {
  const int64_t R = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (R > 3 * N2) return;
  if (R == 3 * N2) { rowptr3[R] = 9 * nadj_ptr[N2]; return; }
  const int64_t r = R / 3;
  const int i = (int)(R % 3);
  const int64_t a = nadj_ptr[r], deg = nadj_ptr[r + 1] - a;
  int64_t pos = 9 * a + 3 * i * deg;
  rowptr3[R] = pos;
  for (int64_t k = 0; k < deg; ++k) {
    const int32_t s = nadj[a + k];
    if (s == r) diagpos3[R] = pos + i;
    cols3[pos++] = 3 * s;
    cols3[pos++] = 3 * s + 1;
    cols3[pos++] = 3 * s + 2;
  }
}
// rows 3r+i x pressure columns (positions in the pressure block) of the vertex neighbours of node r
__global__ void k_vp_structure(int64_t N2, const int64_t* __restrict__ padj_ptr, const int32_t* __restrict__ padj,
                               int64_t* __restrict__ rowptr, int32_t* __restrict__ cols) {
  const int64_t R = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (R > 3 * N2) return;
  if (R == 3 * N2) { rowptr[R] = 3 * padj_ptr[N2]; return; }
  const int64_t r = R / 3;
  const int i = (int)(R % 3);
  const int64_t a = padj_ptr[r], deg = padj_ptr[r + 1] - a;
  int64_t pos = 3 * a + i * deg;
  rowptr[R] = pos;
  for (int64_t k = 0; k < deg; ++k) cols[pos++] = padj[a + k];
}
// pressure rows q x velocity columns 3s+j of the neighbours of the vertex' node; rowptr_pv given (host prefix sum)
__global__ void k_pv_structure(int64_t V, const int32_t* __restrict__ vrank, const int64_t* __restrict__ nadj_ptr,
                               const int32_t* __restrict__ nadj, const int64_t* __restrict__ rowptr,
                               int32_t* __restrict__ cols) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= V) return;
  const int32_t r = vrank[q];
  int64_t pos = rowptr[q];
  for (int64_t k = nadj_ptr[r]; k < nadj_ptr[r + 1]; ++k) {
    const int32_t s = nadj[k];
    cols[pos++] = 3 * s;
    cols[pos++] = 3 * s + 1;
    cols[pos++] = 3 * s + 2;
  }
}

void launch_block_structure(hipStream_t st, int64_t N2, int64_t V, const int64_t* nadj_ptr, const int32_t* nadj,
                            const int64_t* padj_ptr, const int32_t* padj, const int32_t* vrank, int64_t* rowptr3,
                            int32_t* cols3, int64_t* diagpos3, int64_t* rowptr_vp, int32_t* cols_vp,
                            const int64_t* rowptr_pv, int32_t* cols_pv) {
  const unsigned g3 = (unsigned)((3 * N2 + 1 + 255) / 256);
  hipLaunchKernelGGL(k_b3_structure, dim3(g3), dim3(256), 0, st, N2, nadj_ptr, nadj, rowptr3, cols3, diagpos3);
  hipLaunchKernelGGL(k_vp_structure, dim3(g3), dim3(256), 0, st, N2, padj_ptr, padj, rowptr_vp, cols_vp);
  hipLaunchKernelGGL(k_pv_structure, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, vrank, nadj_ptr, nadj,
                     rowptr_pv, cols_pv);
}

// ---- values: split the (row-equilibrated) monolithic matrix into its field blocks -----------------------------------------
// One wave per monolithic row.  Reads nnz*8 B, writes the same once.
__global__ __launch_bounds__(256) void k_extract_blocks(
    int64_t N2, int64_t V, double ktheta, const int64_t* __restrict__ rowptr, const double* __restrict__ A,
    const int64_t* __restrict__ nadj_ptr, const int32_t* __restrict__ nadj, const int64_t* __restrict__ padj_ptr,
    const int32_t* __restrict__ vrank, const int32_t* __restrict__ node_solid, const int64_t* __restrict__ rowptr3,
    const int64_t* __restrict__ rowptr_vp, const int64_t* __restrict__ rowptr_pv, const int64_t* __restrict__ rowptr_pp,
    double* __restrict__ Add, double* __restrict__ Adv, double* __restrict__ Avv, double* __restrict__ Avp,
    double* __restrict__ Apv, double* __restrict__ App) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n = 6 * N2 + V;
  for (int64_t row = wave; row < n; row += nwaves) {
    const int64_t s0 = rowptr[row];
    if (row < 6 * N2) {
      const int64_t r = row / 6;
      const int t = (int)(row % 6), f = t / 3, i = t % 3;
      const int64_t a = nadj_ptr[r], deg = nadj_ptr[r + 1] - a;
      const int64_t o3 = rowptr3[3 * r + i];
      for (int64_t k = lane; k < deg; k += 64) {
        const double* e = A + s0 + 6 * k;
        const double sol = node_solid[nadj[a + k]] ? ktheta : 0.0;
        if (f == 0) {
          for (int j = 0; j < 3; ++j) { Add[o3 + 3 * k + j] = e[j]; Adv[o3 + 3 * k + j] = e[3 + j]; }
        } else {
          for (int j = 0; j < 3; ++j) Avv[o3 + 3 * k + j] = e[3 + j] + sol * e[j];
        }
      }
      if (f == 1) {
        const int64_t pdeg = padj_ptr[r + 1] - padj_ptr[r];
        const int64_t ov = rowptr_vp[3 * r + i];
        for (int64_t k = lane; k < pdeg; k += 64) Avp[ov + k] = A[s0 + 6 * deg + k];
      }
    } else {
      const int64_t q = row - 6 * N2;
      const int32_t r = vrank[q];
      const int64_t a = nadj_ptr[r], deg = nadj_ptr[r + 1] - a;
      const int64_t ov = rowptr_pv[q];
      for (int64_t k = lane; k < deg; k += 64) {
        const double* e = A + s0 + 6 * k;
        const double sol = node_solid[nadj[a + k]] ? ktheta : 0.0;
        for (int j = 0; j < 3; ++j) Apv[ov + 3 * k + j] = e[3 + j] + sol * e[j];
      }
      const int64_t pdeg = padj_ptr[r + 1] - padj_ptr[r];
      const int64_t op = rowptr_pp[q];
      for (int64_t k = lane; k < pdeg; k += 64) App[op + k] = A[s0 + 6 * deg + k];
    }
  }
}
void launch_extract_blocks(hipStream_t st, int64_t N2, int64_t V, double ktheta, const int64_t* rowptr, const double* A,
                           const int64_t* nadj_ptr, const int32_t* nadj, const int64_t* padj_ptr, const int32_t* vrank,
                           const int32_t* node_solid, const int64_t* rowptr3, const int64_t* rowptr_vp,
                           const int64_t* rowptr_pv, const int64_t* rowptr_pp, double* Add, double* Adv, double* Avv,
                           double* Avp, double* Apv, double* App) {
  const int64_t n = 6 * N2 + V;
  int64_t blocks = (n + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_extract_blocks, dim3((unsigned)blocks), dim3(256), 0, st, N2, V, ktheta, rowptr, A, nadj_ptr, nadj,
                     padj_ptr, vrank, node_solid, rowptr3, rowptr_vp, rowptr_pv, rowptr_pp, Add, Adv, Avv, Avp, Apv, App);
}

// ---- explicit Schur complement S = A_pp - Apv~ D^-1 A_vp on its full pattern (vertices within two elements) ----------------
// One wave per pressure row; the pattern (s_rowptr, s_cols: ascending pressure positions) is built once on the host.
static constexpr int MAXS2 = 1024;
__global__ __launch_bounds__(64) void k_schur_full(int64_t V, const int64_t* __restrict__ s_rowptr,
                                                   const int32_t* __restrict__ s_cols, const int32_t* __restrict__ vrank,
                                                   const int64_t* __restrict__ nadj_ptr, const int32_t* __restrict__ nadj,
                                                   const int64_t* __restrict__ padj_ptr, const int32_t* __restrict__ padj,
                                                   const int64_t* __restrict__ rowptr_pv, const double* __restrict__ Apv,
                                                   const int64_t* __restrict__ rowptr_pp, const double* __restrict__ App,
                                                   const int64_t* __restrict__ rowptr_vp, const double* __restrict__ Avp,
                                                   const int64_t* __restrict__ diagpos3, const double* __restrict__ Avv,
                                                   double* __restrict__ S, int32_t* __restrict__ flags) {
  __shared__ double acc[MAXS2];
  __shared__ int32_t scol[MAXS2];
  const int lane = threadIdx.x;
  for (int64_t q = blockIdx.x; q < V; q += gridDim.x) {
    const int64_t s0 = s_rowptr[q];
    const int len = (int)(s_rowptr[q + 1] - s0);
    if (len > MAXS2) { if (lane == 0) atomicOr(&flags[1], 4); continue; }
    __syncthreads();
    for (int t = lane; t < len; t += 64) { acc[t] = 0.0; scol[t] = s_cols[s0 + t]; }
    __syncthreads();
    auto find = [&](int32_t c) {
      int lo = 0, hi = len - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (scol[mid] < c) lo = mid + 1; else hi = mid; }
      return lo;
    };
    const int32_t r = vrank[q];
    {
      const int64_t a = padj_ptr[r], pdeg = padj_ptr[r + 1] - a, op = rowptr_pp[q];
      for (int64_t k = lane; k < pdeg; k += 64) {
        const double v = App[op + k];
        if (v != 0.0) atomicAdd(&acc[find(padj[a + k])], v);
      }
    }
    const int64_t a = nadj_ptr[r], deg = nadj_ptr[r + 1] - a, ov = rowptr_pv[q];
    for (int64_t e = lane; e < 3 * deg; e += 64) {
      const int32_t b = nadj[a + e / 3];
      const int j = (int)(e % 3);
      const double bq = Apv[ov + e];
      if (bq == 0.0) continue;
      const int64_t R = 3 * (int64_t)b + j;
      const double coef = bq / Avv[diagpos3[R]];
      const int64_t pa = padj_ptr[b], pdeg = padj_ptr[b + 1] - pa, o = rowptr_vp[R];
      for (int64_t k = 0; k < pdeg; ++k) {
        const double v = Avp[o + k];
        if (v != 0.0) atomicAdd(&acc[find(padj[pa + k])], -coef * v);
      }
    }
    __syncthreads();
    for (int t = lane; t < len; t += 64) S[s0 + t] = acc[t];
  }
}
void launch_schur_full(hipStream_t st, int64_t V, const int64_t* s_rowptr, const int32_t* s_cols, const int32_t* vrank,
                       const int64_t* nadj_ptr, const int32_t* nadj, const int64_t* padj_ptr, const int32_t* padj,
                       const int64_t* rowptr_pv, const double* Apv, const int64_t* rowptr_pp, const double* App,
                       const int64_t* rowptr_vp, const double* Avp, const int64_t* diagpos3, const double* Avv, double* S,
                       int32_t* flags) {
  const unsigned blocks = (unsigned)(V < 32768 ? V : 32768);
  hipLaunchKernelGGL(k_schur_full, dim3(blocks), dim3(64), 0, st, V, s_rowptr, s_cols, vrank, nadj_ptr, nadj, padj_ptr, padj,
                     rowptr_pv, Apv, rowptr_pp, App, rowptr_vp, Avp, diagpos3, Avv, S, flags);
}

// ---- field vectors <-> monolithic vector (solver ordering: 6 per node [d d d v v v], then pressure) ---------------------------
#define GS(i, n) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)
__global__ void k_split(int64_t N2, int64_t V, const double* __restrict__ r, double* __restrict__ rd,
                        double* __restrict__ rv, double* __restrict__ rp) {
  GS(t, 3 * N2) {
    const int64_t nd = t / 3;
    const int i = (int)(t % 3);
    rd[t] = r[6 * nd + i];
    rv[t] = r[6 * nd + 3 + i];
  }
  GS(q, V) rp[q] = r[6 * N2 + q];
}
__global__ void k_merge(int64_t N2, int64_t V, const double* __restrict__ zd, const double* __restrict__ zv,
                        const double* __restrict__ zp, double* __restrict__ z) {
  GS(t, 3 * N2) {
    const int64_t nd = t / 3;
    const int i = (int)(t % 3);
    z[6 * nd + i] = zd[t];
    z[6 * nd + 3 + i] = zv[t];
  }
  GS(q, V) z[6 * N2 + q] = zp[q];
}
static inline unsigned gridn(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (unsigned)g;
}
void launch_split(hipStream_t st, int64_t N2, int64_t V, const double* r, double* rd, double* rv, double* rp) {
  hipLaunchKernelGGL(k_split, dim3(gridn(3 * N2)), dim3(256), 0, st, N2, V, r, rd, rv, rp);
}
void launch_merge(hipStream_t st, int64_t N2, int64_t V, const double* zd, const double* zv, const double* zp, double* z) {
  hipLaunchKernelGGL(k_merge, dim3(gridn(3 * N2)), dim3(256), 0, st, N2, V, zd, zv, zp, z);
}

// dv[R] = vs[R] - (A_vp dp)[R] / D[R]   (SIMPLE velocity correction); also y = D^-1 A_vp x for the Schur operator.
// One thread per row (rows have 4-30 entries).
// dv = vs - D^-1 A_vp dp.  A row of A_vp has the ~7 vertices its node sees: 8 lanes per row (a wave reads 8 consecutive rows,
// ~56 consecutive entries), sums by DPP, and the update of a workgroup's 32 consecutive rows by its first 32 lanes through
// LDS.  (One thread per row, the first form, walked 4.4 M rows with stride-7 accesses: 375 us, 1 TB/s.)
__global__ __launch_bounds__(256) void k_vel_correct(int64_t n3, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                     const double* __restrict__ vals, const double* __restrict__ dp,
                                                     const int64_t* __restrict__ diagpos3, const double* __restrict__ Avv,
                                                     const double* __restrict__ dinv, const double* __restrict__ vs,
                                                     double* __restrict__ dv) {
  __shared__ double ssum[32];
  const int sub = threadIdx.x & 7, g = threadIdx.x >> 3;
  for (int64_t base = (int64_t)blockIdx.x * 32; base < n3; base += (int64_t)gridDim.x * 32) {
    const int64_t R = base + g;
    double s = 0.0;
    if (R < n3)
      for (int64_t t = rowptr[R] + sub; t < rowptr[R + 1]; t += 8) s += vals[t] * dp[cols[t]];
    s = group_sum<8>(s);
    if (sub == 0) ssum[g] = s;
    __syncthreads();
    if (threadIdx.x < 32 && base + threadIdx.x < n3) {
      const int64_t row = base + threadIdx.x;
      // dinv: 1 / diagonal, made once per refresh (the diagonal itself is a gather from the 1 GB value array of the block)
      dv[row] = (vs ? vs[row] : 0.0) - (dinv ? ssum[threadIdx.x] * dinv[row] : ssum[threadIdx.x] / Avv[diagpos3[row]]);
    }
    __syncthreads();
  }
}
void launch_vel_correct(hipStream_t st, int64_t n3, const int64_t* rowptr, const int32_t* cols, const double* vals,
                        const double* dp, const int64_t* diagpos3, const double* Avv, const double* vs, double* dv, const double* dinv) {
  int64_t blocks = (n3 + 31) / 32;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_vel_correct, dim3((unsigned)blocks), dim3(256), 0, st, n3, rowptr, cols, vals, dp, diagpos3, Avv, dinv, vs, dv);
}
__global__ void k_diag_inverse(int64_t n, const int64_t* __restrict__ diagpos, const double* __restrict__ A, double* __restrict__ dinv) {
  GS(i, n) dinv[i] = 1.0 / A[diagpos[i]];
}
void launch_diag_inverse(hipStream_t st, int64_t n, const int64_t* diagpos, const double* A, double* dinv) {
  hipLaunchKernelGGL(k_diag_inverse, dim3(gridn(n)), dim3(256), 0, st, n, diagpos, A, dinv);
}
// The two block products of the pressure step on FP32 copies of A_vp / A~_pv that use what the block structure gives away:
// the three rows of a node share their vertex columns (padj), a pressure row has three consecutive entries per neighbour
// node (nadj): 16 bytes per (node, vertex) pair instead of 36.  Vectors and sums stay FP64; like every matrix copy of the
// preconditioner these are fixed linear maps.
//   dv = vs - D^-1 A_vp dp        (8 lanes per node, update of a workgroup's 32 nodes = 96 consecutive entries through LDS)
__global__ __launch_bounds__(256) void k_vel_correct32(int64_t N2, const int64_t* __restrict__ padj_ptr, const int32_t* __restrict__ padj,
                                                       const float* __restrict__ avp, const double* __restrict__ dp,
                                                       const double* __restrict__ dinv, const double* __restrict__ vs,
                                                       double* __restrict__ dv) {
  __shared__ double ssum[96];
  const int sub = threadIdx.x & 7, g = threadIdx.x >> 3;
  for (int64_t base = (int64_t)blockIdx.x * 32; base < N2; base += (int64_t)gridDim.x * 32) {
    const int64_t r = base + g;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (r < N2) {
      const int64_t a = padj_ptr[r], deg = padj_ptr[r + 1] - a;
      const float* v = avp + 3 * a;
      for (int64_t k = sub; k < deg; k += 8) {
        const double p = dp[padj[a + k]];
        s0 += (double)v[k] * p; s1 += (double)v[deg + k] * p; s2 += (double)v[2 * deg + k] * p;
      }
    }
    s0 = group_sum<8>(s0); s1 = group_sum<8>(s1); s2 = group_sum<8>(s2);
    if (sub == 0) { ssum[3 * g] = s0; ssum[3 * g + 1] = s1; ssum[3 * g + 2] = s2; }
    __syncthreads();
    if (threadIdx.x < 96 && 3 * base + threadIdx.x < 3 * N2) {
      const int64_t R = 3 * base + threadIdx.x;
      dv[R] = (vs ? vs[R] : 0.0) - ssum[threadIdx.x] * dinv[R];
    }
    __syncthreads();
  }
}
//   y = c - A~_pv w               (16 lanes per pressure row, four strips of neighbours at once)
__global__ __launch_bounds__(256) void k_pres_rhs32(int64_t V, const int32_t* __restrict__ vrank, const int64_t* __restrict__ nadj_ptr,
                                                    const int32_t* __restrict__ nadj, const int64_t* __restrict__ rowptr_pv,
                                                    const float* __restrict__ apv, const double* __restrict__ w,
                                                    const double* __restrict__ c, double* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t q = grp; q < V; q += ngrp) {
    const int32_t r = vrank[q];
    const int64_t a = nadj_ptr[r], deg = nadj_ptr[r + 1] - a;
    const float* v = apv + rowptr_pv[q];
    double s = 0.0;
    for (int64_t k = sub; k < deg; k += 64) {
      int nb[4];
      float f[4][3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = k + 16 * j < deg;
        const int64_t kk = in ? k + 16 * j : k;
        nb[j] = nadj[a + kk];
#pragma unroll
        for (int e = 0; e < 3; ++e) f[j][e] = in ? v[3 * kk + e] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double* ws = w + 3 * (int64_t)nb[j];
        s += (double)f[j][0] * ws[0] + (double)f[j][1] * ws[1] + (double)f[j][2] * ws[2];
      }
    }
    s = group_sum<16>(s);
    if (sub == 0) y[q] = c[q] - s;
  }
}
void launch_vel_correct32(hipStream_t st, int64_t N2, const int64_t* padj_ptr, const int32_t* padj, const float* avp, const double* dp,
                          const double* dinv, const double* vs, double* dv) {
  int64_t blocks = (N2 + 31) / 32;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(k_vel_correct32, dim3((unsigned)blocks), dim3(256), 0, st, N2, padj_ptr, padj, avp, dp, dinv, vs, dv);
}
void launch_pres_rhs32(hipStream_t st, int64_t V, const int32_t* vrank, const int64_t* nadj_ptr, const int32_t* nadj,
                       const int64_t* rowptr_pv, const float* apv, const double* w, const double* c, double* y) {
  int64_t blocks = (V + 15) / 16;
  if (blocks > 32768) blocks = 32768;
  hipLaunchKernelGGL(k_pres_rhs32, dim3((unsigned)blocks), dim3(256), 0, st, V, vrank, nadj_ptr, nadj, rowptr_pv, apv, w, c, y);
}
// y = alpha * (App x)[q] + beta * (Apv~ w)[q] + gamma * c[q]: pressure-row products (Schur operator, pressure rhs).
// One wave per pressure row.
__global__ __launch_bounds__(256) void k_pres_rows(int64_t V, const int64_t* __restrict__ rowptr_pp,
                                                   const int32_t* __restrict__ cols_pp, const double* __restrict__ App,
                                                   const double* __restrict__ x, double alpha,
                                                   const int64_t* __restrict__ rowptr_pv, const int32_t* __restrict__ cols_pv,
                                                   const double* __restrict__ Apv, const double* __restrict__ w, double beta,
                                                   const double* __restrict__ c, double gamma, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t q = wave; q < V; q += nwaves) {
    double s1 = 0.0, s2 = 0.0;
    if (alpha != 0.0)
      for (int64_t t = rowptr_pp[q] + lane; t < rowptr_pp[q + 1]; t += 64) s1 += App[t] * x[cols_pp[t]];
    if (beta != 0.0)
      for (int64_t t = rowptr_pv[q] + lane; t < rowptr_pv[q + 1]; t += 64) s2 += Apv[t] * w[cols_pv[t]];
    const double s = wsum(alpha * s1 + beta * s2);
    if (lane == 0) y[q] = s + (gamma != 0.0 ? gamma * c[q] : 0.0);
  }
}
void launch_pres_rows(hipStream_t st, int64_t V, const int64_t* rowptr_pp, const int32_t* cols_pp, const double* App,
                      const double* x, double alpha, const int64_t* rowptr_pv, const int32_t* cols_pv, const double* Apv,
                      const double* w, double beta, const double* c, double gamma, double* y) {
  int64_t blocks = (V + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_pres_rows, dim3((unsigned)blocks), dim3(256), 0, st, V, rowptr_pp, cols_pp, App, x, alpha, rowptr_pv,
                     cols_pv, Apv, w, beta, c, gamma, y);
}
// ---- Chebyshev semi-iteration on a masked part of the velocity block, Jacobi-scaled -------------------------------------------
// No inner products: a solve is a pure stream of SpMV + one fused vector kernel, nothing returns to the host.
//   init : x = 0, r = mask .* rhs, d = r / (D theta)
//   step : x += d; r -= mask .* t (t = A d); d = c1 d + c2 r / D
__global__ void k_cheb_init(int64_t n, const double* __restrict__ mask, const double* __restrict__ rhs,
                            const int64_t* __restrict__ diagpos, const double* __restrict__ A, double inv_theta,
                            double* __restrict__ x, double* __restrict__ r, double* __restrict__ d) {
  GS(i, n) {
    const double ri = mask ? mask[i] * rhs[i] : rhs[i];
    x[i] = 0.0;
    r[i] = ri;
    d[i] = ri * inv_theta / A[diagpos[i]];
  }
}
__global__ void k_cheb_step(int64_t n, const double* __restrict__ mask, const double* __restrict__ t,
                            const int64_t* __restrict__ diagpos, const double* __restrict__ A, double c1, double c2,
                            double* __restrict__ x, double* __restrict__ r, double* __restrict__ d) {
  GS(i, n) {
    const double di = d[i];
    const double ri = r[i] - (mask ? mask[i] * t[i] : t[i]);
    x[i] += di;
    r[i] = ri;
    d[i] = c1 * di + c2 * ri / A[diagpos[i]];
  }
}
void launch_cheb_init(hipStream_t st, int64_t n, const double* mask, const double* rhs, const int64_t* diagpos,
                      const double* A, double inv_theta, double* x, double* r, double* d) {
  hipLaunchKernelGGL(k_cheb_init, dim3(gridn(n)), dim3(256), 0, st, n, mask, rhs, diagpos, A, inv_theta, x, r, d);
}
void launch_cheb_step(hipStream_t st, int64_t n, const double* mask, const double* t, const int64_t* diagpos,
                      const double* A, double c1, double c2, double* x, double* r, double* d) {
  hipLaunchKernelGGL(k_cheb_step, dim3(gridn(n)), dim3(256), 0, st, n, mask, t, diagpos, A, c1, c2, x, r, d);
}
// y = mask .* (A x) ./ D   (power iteration for the largest eigenvalue of the Jacobi-scaled block)
__global__ void k_mask_scale(int64_t n, const double* __restrict__ mask, const int64_t* __restrict__ diagpos,
                             const double* __restrict__ A, double* __restrict__ y) {
  GS(i, n) y[i] = (mask ? mask[i] : 1.0) * y[i] / A[diagpos[i]];
}
// ---- component-diagonal node-block format ("db"): for node r and neighbour k the three entries (i,i) of the 3x3 block ---------
// A_dd is exactly of this form (mass / mesh Laplacian act per component); the fluid-interior velocity block is dominated
// by it (mass + convection), which is all its Chebyshev *preconditioner* sweeps need.  28 B per node pair instead of 108.
__global__ void k_extract_db(int64_t N2, const int64_t* __restrict__ nadj_ptr, const int64_t* __restrict__ rowptr3,
                             const double* __restrict__ vals, double* __restrict__ db, int32_t* __restrict__ flags, int check) {
  GS(e, nadj_ptr[N2]) {
    // node r of pair e: binary search in nadj_ptr
    int64_t lo = 0, hi = N2 - 1;
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (nadj_ptr[mid] <= e) lo = mid; else hi = mid - 1; }
    const int64_t r = lo, k = e - nadj_ptr[r];
    bool off = false;
    for (int i = 0; i < 3; ++i) {
      const double* row = vals + rowptr3[3 * r + i] + 3 * k;
      db[3 * e + i] = row[i];
      for (int j = 0; j < 3; ++j) off |= (j != i && row[j] != 0.0);
    }
    if (check && off) atomicOr(&flags[1], 8);
  }
}
// y[3r+i] = sum_k db[e][i] x[3 s_k + i]; 16 lanes per node (4 nodes per wave)
// rowmask (may be null): rows whose flag is 0 hold only zeros (A_dv: every fluid row) and are not streamed
__global__ __launch_bounds__(256) void k_spmv_db(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                 const int32_t* __restrict__ nadj, const double* __restrict__ db,
                                                 const uint8_t* __restrict__ rowmask, const double* __restrict__ x,
                                                 double* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = grp; r < N2; r += ngrp) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (!rowmask || rowmask[r]) {
      for (int64_t e = nadj_ptr[r] + sub; e < nadj_ptr[r + 1]; e += 16) {
        const double* xs = x + 3 * (int64_t)nadj[e];
        const double* c = db + 3 * e;
        s0 += c[0] * xs[0]; s1 += c[1] * xs[1]; s2 += c[2] * xs[2];
      }
      s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    }
    if (sub == 0) { y[3 * r] = s0; y[3 * r + 1] = s1; y[3 * r + 2] = s2; }
  }
}
// y[rows of the listed nodes] -= (db x)[those rows]: the A_dv product of the displacement right-hand side, which has entries in
// the rows of solid (incl. interface) nodes only - a tenth of the nodes.  Round 3 ran k_spmv_db over ALL nodes (masked rows
// written as zeros) followed by an axpby over 3 N2 entries: 162 + 22 us per preconditioner application at 1.12 M tets against
// ~20 us for this kernel.  16 lanes per listed node.
__global__ __launch_bounds__(256) void k_db_rows_sub(int64_t nl, const int32_t* __restrict__ list, const int64_t* __restrict__ nadj_ptr,
                                                     const int32_t* __restrict__ nadj, const double* __restrict__ db,
                                                     const uint8_t* __restrict__ rowmask, const double* __restrict__ x,
                                                     double* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t a = grp; a < nl; a += ngrp) {
    const int64_t r = list[a];
    if (rowmask && !rowmask[r]) continue;            // uniform over the 16 lanes of the node
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t e = nadj_ptr[r] + sub; e < nadj_ptr[r + 1]; e += 16) {
      const double* xs = x + 3 * (int64_t)nadj[e];
      const double* c = db + 3 * e;
      s0 += c[0] * xs[0]; s1 += c[1] * xs[1]; s2 += c[2] * xs[2];
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) { y[3 * r] -= s0; y[3 * r + 1] -= s1; y[3 * r + 2] -= s2; }
  }
}
// flags[0] |= 1 if a node outside the list's set (node_set[r] == 0) has rowmask[r] set
__global__ void k_mask_outside(int64_t N2, const uint8_t* __restrict__ rowmask, const int32_t* __restrict__ node_set, int32_t* __restrict__ flags) {
  GS(r, N2) if (rowmask[r] && !node_set[r]) atomicOr(&flags[0], 1);
}
void launch_mask_outside(hipStream_t st, int64_t N2, const uint8_t* rowmask, const int32_t* node_set, int32_t* flags) {
  hipLaunchKernelGGL(k_mask_outside, dim3(gridn(N2)), dim3(256), 0, st, N2, rowmask, node_set, flags);
}
void launch_db_rows_sub(hipStream_t st, int64_t nl, const int32_t* list, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                        const uint8_t* rowmask, const double* x, double* y) {
  if (nl <= 0) return;
  int64_t blocks = (nl + 15) / 16;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_db_rows_sub, dim3((unsigned)blocks), dim3(256), 0, st, nl, list, nadj_ptr, nadj, db, rowmask, x, y);
}
// rowmask[r] = 1 iff node r has a non-zero entry in db
__global__ __launch_bounds__(256) void k_db_rowmask(int64_t N2, const int64_t* __restrict__ nadj_ptr, const double* __restrict__ db,
                                                    uint8_t* __restrict__ rowmask) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = grp; r < N2; r += ngrp) {
    float any = 0.f;
    for (int64_t e = nadj_ptr[r] + sub; e < nadj_ptr[r + 1]; e += 16)
      if (db[3 * e] != 0.0 || db[3 * e + 1] != 0.0 || db[3 * e + 2] != 0.0) any = 1.f;
    any = group_sum<16>(any);
    if (sub == 0) rowmask[r] = any > 0.f ? 1 : 0;
  }
}
void launch_db_rowmask(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const double* db, uint8_t* rowmask) {
  int64_t blocks = (N2 + 15) / 16;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_db_rowmask, dim3((unsigned)blocks), dim3(256), 0, st, N2, nadj_ptr, db, rowmask);
}
// FP32 variants for the preconditioner sweeps (values converted once per Jacobian refresh)
__global__ __launch_bounds__(256) void k_spmv_db_f32(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                     const int32_t* __restrict__ nadj, const float* __restrict__ db,
                                                     const float* __restrict__ x, float* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = grp; r < N2; r += ngrp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t e = nadj_ptr[r] + sub; e < nadj_ptr[r + 1]; e += 16) {
      const float4 xv = reinterpret_cast<const float4*>(x)[nadj[e]];
      const float* c = db + 3 * e;
      s0 += c[0] * xv.x; s1 += c[1] * xv.y; s2 += c[2] * xv.z;
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) reinterpret_cast<float4*>(y)[r] = make_float4(s0, s1, s2, 0.f);
  }
}
__global__ void k_to_f32(int64_t n, const double* __restrict__ a, float* __restrict__ b) { GS(i, n) b[i] = (float)a[i]; }
// node vectors: [3 per node] double  <->  [4 per node] float (pad = 0)
__global__ void k_pad_to_f32(int64_t nn, const double* __restrict__ a, const float* __restrict__ scale4, float* __restrict__ b) {
  GS(t, 4 * nn) {
    const int64_t nd = t >> 2;
    const int c = (int)(t & 3);
    b[t] = c < 3 ? (float)a[3 * nd + c] * (scale4 ? scale4[t] : 1.f) : 0.f;
  }
}
__global__ void k_unpad_from_f32(int64_t nn, const float* __restrict__ a, double* __restrict__ b) {
  GS(t, 3 * nn) b[t] = (double)a[4 * (t / 3) + t % 3];
}
// pad_to_f32 + cheb_init_f32 in one launch: rhs = a (3 doubles per node) [* scale4], x = 0, r = rhs, d = rhs inv_theta dinv4
__global__ void k_pad_init_f32(int64_t nn, const double* __restrict__ a, const float* __restrict__ scale4, const float* __restrict__ dinv4,
                               float inv_theta, float* __restrict__ x, float* __restrict__ r, float* __restrict__ d) {
  GS(t, 4 * nn) {
    const int64_t nd = t >> 2;
    const int c = (int)(t & 3);
    const float ri = c < 3 ? (float)a[3 * nd + c] * (scale4 ? scale4[t] : 1.f) : 0.f;
    x[t] = 0.f; r[t] = ri; d[t] = ri * inv_theta * dinv4[t];
  }
}
// merge with the displacement part taken straight from the sweeps' float4 result (unpad_from_f32 + merge in one launch)
__global__ void k_merge_f32d(int64_t N2, int64_t V, const float* __restrict__ xd4, const double* __restrict__ zv,
                             const double* __restrict__ zp, double* __restrict__ z) {
  GS(t, 3 * N2) {
    const int64_t nd = t / 3;
    const int i = (int)(t % 3);
    z[6 * nd + i] = (double)xd4[4 * nd + i];
    z[6 * nd + 3 + i] = zv[t];
  }
  GS(q, V) z[6 * N2 + q] = zp[q];
}
void launch_pad_init_f32(hipStream_t st, int64_t nn, const double* a, const float* scale4, const float* dinv4, float inv_theta,
                         float* x, float* r, float* d) {
  hipLaunchKernelGGL(k_pad_init_f32, dim3(gridn(4 * nn)), dim3(256), 0, st, nn, a, scale4, dinv4, inv_theta, x, r, d);
}
void launch_merge_f32d(hipStream_t st, int64_t N2, int64_t V, const float* xd4, const double* zv, const double* zp, double* z) {
  hipLaunchKernelGGL(k_merge_f32d, dim3(gridn(3 * N2)), dim3(256), 0, st, N2, V, xd4, zv, zp, z);
}
// dinv4[4 nd + c] = mask / A[diagpos[3 nd + c]], pad 0  (mask may be null; one4: write 1 instead of the inverse diagonal)
__global__ void k_dinv_f32(int64_t nn, const double* __restrict__ mask, const int64_t* __restrict__ diagpos,
                           const double* __restrict__ A, float* __restrict__ dinv4) {
  GS(t, 4 * nn) {
    const int64_t nd = t >> 2;
    const int c = (int)(t & 3);
    dinv4[t] = c < 3 ? (float)((mask ? mask[3 * nd + c] : 1.0) / A[diagpos[3 * nd + c]]) : 0.f;
  }
}
// A_dd is (scalar node-pair matrix) x I_3 up to the row scaling and the Dirichlet rows; its Jacobi-scaled form
// D^-1 A_dd therefore needs ONE number per node pair, chat = c_rs / c_rr, plus a flag per Dirichlet row.
__global__ void k_extract_chat(int64_t N2, const int64_t* __restrict__ nadj_ptr, const int32_t* __restrict__ nadj,
                               const double* __restrict__ db, float* __restrict__ chat, uint8_t* __restrict__ rowflag,
                               int32_t* __restrict__ flags) {
  GS(r, N2) {
    const int64_t a = nadj_ptr[r], b = nadj_ptr[r + 1];
    int64_t ed = -1;
    for (int64_t e = a; e < b; ++e) if (nadj[e] == r) ed = e;
    // a row is "identity" (Dirichlet / ident_zeros) when its only entry is the diagonal
    bool ident[3];
    for (int i = 0; i < 3; ++i) {
      bool off = false;
      for (int64_t e = a; e < b; ++e) off |= (e != ed && db[3 * e + i] != 0.0);
      ident[i] = !off;
      rowflag[3 * r + i] = ident[i] ? 1 : 0;
    }
    int ref = -1;
    for (int i = 0; i < 3; ++i) if (!ident[i] && ref < 0) ref = i;
    for (int64_t e = a; e < b; ++e) {
      if (ref < 0) { chat[e] = (e == ed) ? 1.f : 0.f; continue; }
      const double c = db[3 * e + ref] / db[3 * ed + ref];
      chat[e] = (float)c;
      for (int i = 0; i < 3; ++i)       // the other free components must carry the same ratios
        if (!ident[i] && fabs(db[3 * e + i] / db[3 * ed + i] - c) > 1e-9 * (fabs(c) + 1e-30) + 1e-12) atomicOr(&flags[1], 16);
    }
  }
}
__global__ __launch_bounds__(256) void k_spmv_sc_f32(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                     const int32_t* __restrict__ nadj, const float* __restrict__ chat,
                                                     const uint8_t* __restrict__ rowflag, const float* __restrict__ x,
                                                     float* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = grp; r < N2; r += ngrp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t e = nadj_ptr[r] + sub; e < nadj_ptr[r + 1]; e += 16) {
      const float4 xv = reinterpret_cast<const float4*>(x)[nadj[e]];
      const float c = chat[e];
      s0 += c * xv.x; s1 += c * xv.y; s2 += c * xv.z;
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) {
      const float4 xr = reinterpret_cast<const float4*>(x)[r];
      reinterpret_cast<float4*>(y)[r] = make_float4(rowflag[3 * r] ? xr.x : s0, rowflag[3 * r + 1] ? xr.y : s1,
                                                    rowflag[3 * r + 2] ? xr.z : s2, 0.f);
    }
  }
}

// ---- LDS-tiled variants: a workgroup owns TN (128 | 256, per context) consecutive nodes; the vector entries of all their neighbours
// (a few thousand distinct nodes in mesh order) are gathered ONCE into LDS, the node-pair loop then reads 16-byte
// entries from LDS through 2-byte local indices.  HBM per pair: value(s) + 2 B instead of value(s) + 4 B + a 16-B
// gather that the L1/L2 have to serve.
// The tile of a workgroup: in the XCD-aware order of fsi_kernels.hpp (grid = xcd_grid(tiles); -1: no tile)
__device__ inline int64_t xcd_tile(int64_t ntiles) { return xcd_unit(blockIdx.x, ntiles); }
static constexpr int TILE_LIMIT = 3584;          // distinct neighbour nodes per tile (56 KB of LDS as float4; + 6 KB static < 64 KB per workgroup)
template <int NV, int TN>     // NV = 1: one ratio per pair (displacement block), NV = 3: component-diagonal values (fluid velocity block)
__global__ __launch_bounds__(256) void k_spmv_tiled_f32(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                        const float* __restrict__ vals, const uint16_t* __restrict__ ploc,
                                                        const int64_t* __restrict__ tile_uptr, const int32_t* __restrict__ ulist,
                                                        const uint8_t* __restrict__ rowflag, const float* __restrict__ x,
                                                        float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float4 sx[];   // max over the tiles of their distinct-neighbour count
  __shared__ __attribute__((aligned(16))) int64_t sptr[TN + 2];   // size a multiple of 16 B: keeps the dynamic base aligned
  const int64_t tile = xcd_tile((N2 + TN - 1) / TN);
  if (tile < 0) return;
  const int64_t u0 = tile_uptr[tile], nu = tile_uptr[tile + 1] - u0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const int64_t r0 = tile * TN;
  const int nrows = (int)((r0 + TN < N2 ? r0 + TN : N2) - r0);
  for (int64_t i = threadIdx.x; i < nu; i += 256) sx[i] = x4[ulist[u0 + i]];
  for (int i = threadIdx.x; i <= nrows; i += 256) sptr[i] = nadj_ptr[r0 + i];
  __syncthreads();
  const int sub = threadIdx.x & 15, g = threadIdx.x >> 4;
  // the node loop is software-pipelined: the (value, local index) pairs of the group's NEXT node are in flight while
  // the current node is multiplied and reduced - the loop is bound by load latency, not by bytes
  float cv[2][NV], nv_[2][NV];
  int cl[2], nl[2];
  auto prefetch = [&](int i, float (&v)[2][NV], int (&l)[2]) {
    const int64_t e0 = sptr[i], e1 = sptr[i + 1];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t e = e0 + sub + 16 * k;
      const bool in = e < e1;
      l[k] = in ? (int)ploc[e] : 0;
#pragma unroll
      for (int c = 0; c < NV; ++c) v[k][c] = in ? vals[NV * e + c] : 0.f;
    }
  };
  int i = g;
  if (i < nrows) prefetch(i, cv, cl);
  while (i < nrows) {
    const int ni = i + 16;
    if (ni < nrows) prefetch(ni, nv_, nl);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 xv = sx[cl[k]];
      if (NV == 1) { s0 += cv[k][0] * xv.x; s1 += cv[k][0] * xv.y; s2 += cv[k][0] * xv.z; }
      else { s0 += cv[k][0] * xv.x; s1 += cv[k][NV > 1 ? 1 : 0] * xv.y; s2 += cv[k][NV > 2 ? 2 : 0] * xv.z; }
    }
    for (int64_t e = sptr[i] + sub + 32; e < sptr[i + 1]; e += 16) {      // rows with more than 32 pairs
      const float4 xv = sx[ploc[e]];
      if (NV == 1) { const float c = vals[e]; s0 += c * xv.x; s1 += c * xv.y; s2 += c * xv.z; }
      else { const float* c = vals + NV * e; s0 += c[0] * xv.x; s1 += c[NV > 1 ? 1 : 0] * xv.y; s2 += c[NV > 2 ? 2 : 0] * xv.z; }
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) {
      const int64_t r = r0 + i;
      if (rowflag) {
        const float4 xr = x4[r];
        reinterpret_cast<float4*>(y)[r] = make_float4(rowflag[3 * r] ? xr.x : s0, rowflag[3 * r + 1] ? xr.y : s1,
                                                      rowflag[3 * r + 2] ? xr.z : s2, 0.f);
      } else {
        reinterpret_cast<float4*>(y)[r] = make_float4(s0, s1, s2, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      cl[k] = nl[k];
#pragma unroll
      for (int c = 0; c < NV; ++c) cv[k][c] = nv_[k][c];
    }
    i = ni;
  }
}
// (the launchers take the context's tile size tn = 128 | 256: FsiTuning.tile_nodes, 0 = by the number of nodes)
#define TILED_DISPATCH(KERNEL, THREADS, ...)                                                                                     \
  do {                                                                                                                           \
    if (nv == 1 && tn == 128) hipLaunchKernelGGL((KERNEL<1, 128>), dim3(tiles), dim3(THREADS), lds, st, __VA_ARGS__);           \
    else if (nv == 1) hipLaunchKernelGGL((KERNEL<1, 256>), dim3(tiles), dim3(THREADS), lds, st, __VA_ARGS__);                   \
    else if (tn == 128) hipLaunchKernelGGL((KERNEL<3, 128>), dim3(tiles), dim3(THREADS), lds, st, __VA_ARGS__);                 \
    else hipLaunchKernelGGL((KERNEL<3, 256>), dim3(tiles), dim3(THREADS), lds, st, __VA_ARGS__);                                \
  } while (0)
void launch_spmv_tiled_f32(hipStream_t st, int nv, int tn, int64_t N2, int max_nu, const int64_t* nadj_ptr, const float* vals,
                           const uint16_t* ploc, const int64_t* tile_uptr, const int32_t* ulist, const uint8_t* rowflag,
                           const float* x, float* y) {
  const unsigned tiles = xcd_grid((N2 + tn - 1) / tn);
  const size_t lds = (size_t)max_nu * sizeof(float4);
  TILED_DISPATCH(k_spmv_tiled_f32, 256, N2, nadj_ptr, vals, ploc, tile_uptr, ulist, rowflag, x, y);
}
// One Chebyshev sweep on the tiled operator in a single launch (the product never goes through memory):
//   t = A d_in;  x += d_in;  r -= t;  d_out = c1 d_in + c2 dinv r      (d is ping-ponged: other tiles still gather d_in)
// Workgroups of up to 1024 threads: the tile's LDS (up to 64 KB) allows two workgroups per CU whatever their size, and the
// row loop is bound by load latency, so 32 waves per CU instead of 8 is what the kernel is after.
template <int NV, int TN>
__global__ __launch_bounds__(1024) void k_sweep_tiled_f32(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                          const float* __restrict__ vals, const uint16_t* __restrict__ ploc,
                                                          const int64_t* __restrict__ tile_uptr, const int32_t* __restrict__ ulist,
                                                          const uint8_t* __restrict__ rowflag, const float* __restrict__ dinv,
                                                          float c1, float c2, const float* __restrict__ din, float* __restrict__ dout,
                                                          float* __restrict__ x, float* __restrict__ r) {
  extern __shared__ __attribute__((aligned(16))) float4 sx[];
  __shared__ __attribute__((aligned(16))) int64_t sptr[TN + 2];
  __shared__ __attribute__((aligned(16))) float4 ssum[TN];   // the tile's products; the update below reads them coalesced
  const int64_t tile = xcd_tile((N2 + TN - 1) / TN);
  if (tile < 0) return;
  const int64_t u0 = tile_uptr[tile], nu = tile_uptr[tile + 1] - u0;
  const float4* d4 = reinterpret_cast<const float4*>(din);
  const int64_t r0 = tile * TN;
  const int nrows = (int)((r0 + TN < N2 ? r0 + TN : N2) - r0);
  const int nth = blockDim.x, ngrp = nth >> 4;
  for (int64_t i = threadIdx.x; i < nu; i += 4 * nth) {        // index -> entry is a dependent pair of loads: four pairs in flight
    const int64_t i1 = i + nth, i2 = i + 2 * nth, i3 = i + 3 * nth;
    const int32_t k0 = ulist[u0 + i], k1 = i1 < nu ? ulist[u0 + i1] : 0, k2 = i2 < nu ? ulist[u0 + i2] : 0, k3 = i3 < nu ? ulist[u0 + i3] : 0;
    const float4 v0 = d4[k0], v1 = d4[k1], v2 = d4[k2], v3 = d4[k3];
    sx[i] = v0;
    if (i1 < nu) sx[i1] = v1;
    if (i2 < nu) sx[i2] = v2;
    if (i3 < nu) sx[i3] = v3;
  }
  for (int i = threadIdx.x; i <= nrows; i += nth) sptr[i] = nadj_ptr[r0 + i];
  __syncthreads();
  const int sub = threadIdx.x & 15, g = threadIdx.x >> 4;
  // The (value, local index) pairs of the group's NEXT node are in flight while the current node is multiplied and reduced.
  // Four 16-pair strips are prefetched: a P2 edge node has ~22 neighbours, a vertex node ~65.  (Measured alternatives, all
  // slower on MI355X: two strips + a loop of dependent loads for the long rows, 184 us; the rows of a tile sorted into long
  // and short ones with five / two strips, 152 us; this form 142 us.)
  constexpr int KS = 4;
  float cv[KS][NV], nv_[KS][NV];
  int cl[KS], nl[KS];
  auto prefetch = [&](int i, float (&v)[KS][NV], int (&l)[KS]) {
    const int64_t e0 = sptr[i], e1 = sptr[i + 1];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int64_t e = e0 + sub + 16 * k;
      const bool in = e < e1;
      l[k] = in ? (int)ploc[e] : 0;
#pragma unroll
      for (int c = 0; c < NV; ++c) v[k][c] = in ? vals[NV * e + c] : 0.f;
    }
  };
  int i = g;
  if (i < nrows) prefetch(i, cv, cl);
  while (i < nrows) {
    const int ni = i + ngrp;
    if (ni < nrows) prefetch(ni, nv_, nl);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    auto strip = [&](int k) {
      const float4 xv = sx[cl[k]];
      if (NV == 1) { s0 += cv[k][0] * xv.x; s1 += cv[k][0] * xv.y; s2 += cv[k][0] * xv.z; }
      else { s0 += cv[k][0] * xv.x; s1 += cv[k][NV > 1 ? 1 : 0] * xv.y; s2 += cv[k][NV > 2 ? 2 : 0] * xv.z; }
    };
    strip(0);
    strip(1);
    const int len = (int)(sptr[i + 1] - sptr[i]);
    if (__builtin_amdgcn_ballot_w64(len > 32) != 0) { strip(2); strip(3); }        // wave-uniform: some node of the wave is long
    for (int64_t e = sptr[i] + sub + 16 * KS; e < sptr[i + 1]; e += 16) {           // rows with more than 64 pairs
      const float4 xv = sx[ploc[e]];
      if (NV == 1) { const float c = vals[e]; s0 += c * xv.x; s1 += c * xv.y; s2 += c * xv.z; }
      else { const float* c = vals + NV * e; s0 += c[0] * xv.x; s1 += c[NV > 1 ? 1 : 0] * xv.y; s2 += c[NV > 2 ? 2 : 0] * xv.z; }
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) ssum[i] = make_float4(s0, s1, s2, 0.f);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      cl[k] = nl[k];
#pragma unroll
      for (int c = 0; c < NV; ++c) cv[k][c] = nv_[k][c];
    }
    i = ni;
  }
  // Chebyshev update of the tile's rows, one float per thread and round: the tile's entries of d, r, x are 4 KB of
  // consecutive memory, read and written by full waves (done per row by the lane that holds the sum, these nine accesses
  // with 4 of 64 lanes active cost more address-unit cycles than the product itself)
  __syncthreads();
  const float* sflat = reinterpret_cast<const float*>(ssum);
  for (int idx = threadIdx.x; idx < 4 * nrows; idx += nth) {
    const int64_t gi = 4 * r0 + idx;
    const int comp = idx & 3;
    const float di = din[gi];
    float t = sflat[idx];
    if (rowflag && comp < 3 && rowflag[3 * (r0 + (idx >> 2)) + comp]) t = di;      // identity (Dirichlet) rows of the scaled operator
    const float ri = r[gi] - t;
    x[gi] += di;
    r[gi] = ri;
    dout[gi] = comp < 3 ? c1 * di + c2 * ri * (dinv ? dinv[gi] : 1.f) : 0.f;
  }
}
// (The dynamic LDS is sized for the largest tile; on the 1.12 M-tet mesh the tiles gather 889 distinct neighbours in the
// median and 1325 at most, 21 KB, so LDS does not limit the occupancy.  Splitting the tiles into two launches by size was
// measured: slower, 159 against 143 us.)
void launch_sweep_tiled_f32(hipStream_t st, int nv, int tn, int64_t N2, int max_nu, const int64_t* nadj_ptr, const float* vals,
                            const uint16_t* ploc, const int64_t* tile_uptr, const int32_t* ulist, const uint8_t* rowflag,
                            const float* dinv, float c1, float c2, const float* din, float* dout, float* x, float* r) {
  const int th = tn == 128 ? 256 : 512;      // 16 lanes per node, 8 rounds per tile (measured at 256 nodes: 1024 threads no gain over 512)
  const unsigned tiles = xcd_grid((N2 + tn - 1) / tn);
  const size_t lds = (size_t)max_nu * sizeof(float4);
  TILED_DISPATCH(k_sweep_tiled_f32, th, N2, nadj_ptr, vals, ploc, tile_uptr, ulist, rowflag, dinv, c1, c2, din, dout, x, r);
}
// ---- FP16 matrix values for the fine-level sweeps ---------------------------------------------------------------------
// The sweeps are a fixed polynomial in a matrix that only has to resemble the block it preconditions; rounding its VALUES
// to FP16 (11 bits; Jacobi-scaled ratios and row-equilibrated entries, |.| <= 1) changes the preconditioner by 5e-4 of
// itself - nothing the outer iteration can see - while every vector stays FP32 / FP64.  What it buys is the record: value
// and 16-bit local column index of a node pair in ONE 4-byte load (NV = 1; 6 bytes and two loads before), three values and
// the index in ONE 8-byte load (NV = 3; 14 bytes and four loads), a 3x3 solid block and its column in three 8-byte loads
// (24 bytes; 40 bytes and ten loads).  Fewer bytes and, what paces these kernels as much, fewer memory instructions.
__device__ inline float h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits16); }
__device__ inline uint32_t f2h(float v) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v); }
__global__ void k_pack_h1(int64_t n, const float* __restrict__ v, const uint16_t* __restrict__ loc, uint32_t* __restrict__ rec) {
  GS(e, n) rec[e] = f2h(v[e]) | ((uint32_t)loc[e] << 16);
}
__global__ void k_pack_h3(int64_t n, const float* __restrict__ v, const uint16_t* __restrict__ loc, uint2* __restrict__ rec) {
  GS(e, n) rec[e] = make_uint2(f2h(v[3 * e]) | (f2h(v[3 * e + 1]) << 16), f2h(v[3 * e + 2]) | ((uint32_t)loc[e] << 16));
}
// rec[3 b + 0..2]: (a0 a1 | a2 a3), (a4 a5 | a6 a7), (a8 0 | column)
__global__ void k_pack_sb(int64_t nb, const float* __restrict__ v, const int32_t* __restrict__ col, uint2* __restrict__ rec) {
  GS(b, nb) {
    const float* a = v + 9 * b;
    rec[3 * b] = make_uint2(f2h(a[0]) | (f2h(a[1]) << 16), f2h(a[2]) | (f2h(a[3]) << 16));
    rec[3 * b + 1] = make_uint2(f2h(a[4]) | (f2h(a[5]) << 16), f2h(a[6]) | (f2h(a[7]) << 16));
    rec[3 * b + 2] = make_uint2(f2h(a[8]), (uint32_t)col[b]);
  }
}
void launch_pack_h1(hipStream_t st, int64_t n, const float* v, const uint16_t* loc, uint32_t* rec) {
  hipLaunchKernelGGL(k_pack_h1, dim3(gridn(n)), dim3(256), 0, st, n, v, loc, rec);
}
void launch_pack_h3(hipStream_t st, int64_t n, const float* v, const uint16_t* loc, void* rec) {
  hipLaunchKernelGGL(k_pack_h3, dim3(gridn(n)), dim3(256), 0, st, n, v, loc, static_cast<uint2*>(rec));
}
void launch_pack_sb(hipStream_t st, int64_t nb, const float* v, const int32_t* col, void* rec) {
  hipLaunchKernelGGL(k_pack_sb, dim3(gridn(nb)), dim3(256), 0, st, nb, v, col, static_cast<uint2*>(rec));
}
template <int NV> struct TileRec;
template <> struct TileRec<1> { using type = uint32_t; };
template <> struct TileRec<3> { using type = uint2; };
// k_sweep_tiled_f32 on the packed records
template <int NV, int TN>
__global__ __launch_bounds__(1024) void k_sweep_tiled_h(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                        const typename TileRec<NV>::type* __restrict__ rec,
                                                        const int64_t* __restrict__ tile_uptr, const int32_t* __restrict__ ulist,
                                                        const uint8_t* __restrict__ rowflag, const float* __restrict__ dinv,
                                                        float c1, float c2, const float* __restrict__ din, float* __restrict__ dout,
                                                        float* __restrict__ x, float* __restrict__ r) {
  using Rec = typename TileRec<NV>::type;
  extern __shared__ __attribute__((aligned(16))) float4 sx[];
  __shared__ __attribute__((aligned(16))) int64_t sptr[TN + 2];
  __shared__ __attribute__((aligned(16))) float4 ssum[TN];
  const int64_t tile = xcd_tile((N2 + TN - 1) / TN);
  if (tile < 0) return;
  const int64_t u0 = tile_uptr[tile], nu = tile_uptr[tile + 1] - u0;
  const float4* d4 = reinterpret_cast<const float4*>(din);
  const int64_t r0 = tile * TN;
  const int nrows = (int)((r0 + TN < N2 ? r0 + TN : N2) - r0);
  const int nth = blockDim.x, ngrp = nth >> 4;
  for (int64_t i = threadIdx.x; i < nu; i += 4 * nth) {        // index -> entry is a dependent pair of loads: four pairs in flight
    const int64_t i1 = i + nth, i2 = i + 2 * nth, i3 = i + 3 * nth;
    const int32_t k0 = ulist[u0 + i], k1 = i1 < nu ? ulist[u0 + i1] : 0, k2 = i2 < nu ? ulist[u0 + i2] : 0, k3 = i3 < nu ? ulist[u0 + i3] : 0;
    const float4 v0 = d4[k0], v1 = d4[k1], v2 = d4[k2], v3 = d4[k3];
    sx[i] = v0;
    if (i1 < nu) sx[i1] = v1;
    if (i2 < nu) sx[i2] = v2;
    if (i3 < nu) sx[i3] = v3;
  }
  for (int i = threadIdx.x; i <= nrows; i += nth) sptr[i] = nadj_ptr[r0 + i];
  __syncthreads();
  const int sub = threadIdx.x & 15, g = threadIdx.x >> 4;
  constexpr int KS = 4;
  Rec cr[KS], nr[KS];
  auto zero = [](Rec& q) { if constexpr (NV == 1) q = 0u; else q = make_uint2(0u, 0u); };
  auto prefetch = [&](int i, Rec (&q)[KS]) {
    const int64_t e0 = sptr[i], e1 = sptr[i + 1];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int64_t e = e0 + sub + 16 * k;
      if (e < e1) q[k] = rec[e]; else zero(q[k]);               // a zero record: value +0, local index 0
    }
  };
  auto fma3 = [&](const Rec q, float& s0, float& s1, float& s2) {
    if constexpr (NV == 1) {
      const float c = h2f(q & 0xffffu);
      const float4 xv = sx[q >> 16];
      s0 += c * xv.x; s1 += c * xv.y; s2 += c * xv.z;
    } else {
      const float4 xv = sx[q.y >> 16];
      s0 += h2f(q.x & 0xffffu) * xv.x; s1 += h2f(q.x >> 16) * xv.y; s2 += h2f(q.y & 0xffffu) * xv.z;
    }
  };
  // The records of TWO rows ahead are in flight while a row is summed: a row is ~28 pairs of 4 - 8 bytes over 16 lanes, and one row
  // ahead left ~32 KB in flight per CU, which paced the one-ratio form at ~3.7 TB/s (119.9 -> 112.1 us per displacement sweep at
  // 1.12 M tets, A / B on one box; the 8-byte records of the fluid block: 144 us either way).
  int i = g;
  Rec n2[KS];
  if (i < nrows) prefetch(i, cr);
  if (i + ngrp < nrows) prefetch(i + ngrp, nr);
  while (i < nrows) {
    const int ni = i + ngrp;
    if (ni + ngrp < nrows) prefetch(ni + ngrp, n2);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    fma3(cr[0], s0, s1, s2);
    fma3(cr[1], s0, s1, s2);
    const int len = (int)(sptr[i + 1] - sptr[i]);
    if (__builtin_amdgcn_ballot_w64(len > 32) != 0) { fma3(cr[2], s0, s1, s2); fma3(cr[3], s0, s1, s2); }
    for (int64_t e = sptr[i] + sub + 16 * KS; e < sptr[i + 1]; e += 16) fma3(rec[e], s0, s1, s2);      // more than 64 pairs
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) ssum[i] = make_float4(s0, s1, s2, 0.f);
#pragma unroll
    for (int k = 0; k < KS; ++k) { cr[k] = nr[k]; nr[k] = n2[k]; }
    i = ni;
  }
  __syncthreads();
  const float* sflat = reinterpret_cast<const float*>(ssum);
  for (int idx = threadIdx.x; idx < 4 * nrows; idx += nth) {
    const int64_t gi = 4 * r0 + idx;
    const int comp = idx & 3;
    const float di = din[gi];
    float t = sflat[idx];
    if (rowflag && comp < 3 && rowflag[3 * (r0 + (idx >> 2)) + comp]) t = di;
    const float ri = r[gi] - t;
    x[gi] += di;
    r[gi] = ri;
    dout[gi] = comp < 3 ? c1 * di + c2 * ri * (dinv ? dinv[gi] : 1.f) : 0.f;
  }
}
void launch_sweep_tiled_h(hipStream_t st, int nv, int tn, int64_t N2, int max_nu, const int64_t* nadj_ptr, const void* rec,
                          const int64_t* tile_uptr, const int32_t* ulist, const uint8_t* rowflag, const float* dinv, float c1,
                          float c2, const float* din, float* dout, float* x, float* r) {
  const int th = tn == 128 ? 256 : 512;
  const unsigned tiles = xcd_grid((N2 + tn - 1) / tn);
  const size_t lds = (size_t)max_nu * sizeof(float4);
  if (nv == 1 && tn == 128)
    hipLaunchKernelGGL((k_sweep_tiled_h<1, 128>), dim3(tiles), dim3(th), lds, st, N2, nadj_ptr, static_cast<const uint32_t*>(rec), tile_uptr, ulist, rowflag, dinv, c1, c2, din, dout, x, r);
  else if (nv == 1)
    hipLaunchKernelGGL((k_sweep_tiled_h<1, 256>), dim3(tiles), dim3(th), lds, st, N2, nadj_ptr, static_cast<const uint32_t*>(rec), tile_uptr, ulist, rowflag, dinv, c1, c2, din, dout, x, r);
  else if (tn == 128)
    hipLaunchKernelGGL((k_sweep_tiled_h<3, 128>), dim3(tiles), dim3(th), lds, st, N2, nadj_ptr, static_cast<const uint2*>(rec), tile_uptr, ulist, rowflag, dinv, c1, c2, din, dout, x, r);
  else
    hipLaunchKernelGGL((k_sweep_tiled_h<3, 256>), dim3(tiles), dim3(th), lds, st, N2, nadj_ptr, static_cast<const uint2*>(rec), tile_uptr, ulist, rowflag, dinv, c1, c2, din, dout, x, r);
}
// k_sweep_sb_b3<0> on the packed 24-byte block records.  LPR lanes per row with 32 / LPR blocks in flight per lane: the sweep is a chain
// of three dependent loads (row pointer -> record -> gathered d) per round of resident waves, and 150 k rows of 16 lanes are 4.6 rounds;
// 8 lanes per row are half the rounds with twice the loads in flight per lane (measured at 1.12 M tets, A / B on one box: 16 lanes
// 25.7 us per sweep - 28.4 before the loads of a strip were issued together -, 8 lanes 23.8 us, 4 lanes 33.4 us).
template <int LPR>
__global__ __launch_bounds__(256) void k_sweep_sb_h(int64_t nS, const int64_t* __restrict__ sb_ptr, const uint2* __restrict__ rec,
                                                    const float* __restrict__ binv12, float c1, float c2,
                                                    const float* __restrict__ din, float* __restrict__ dout,
                                                    float* __restrict__ x, float* __restrict__ r) {
  constexpr int NB = 32 / LPR;                               // blocks in flight per lane
  const int sub = threadIdx.x & (LPR - 1);
  // rows in the workgroups' XCD-aware order (xcd_tile): a workgroup's 256 / LPR rows gather the d entries of rows a few workgroups away
  const int64_t lb = xcd_tile((nS * LPR + 255) / 256);
  if (lb < 0) return;
  const int64_t grp = (lb * (int64_t)blockDim.x + threadIdx.x) / LPR;
  const int64_t ngrp = nS;                                   // one row per group: the grid covers the rows
  const float4* d4 = reinterpret_cast<const float4*>(din);
  for (int64_t i = grp; i < nS; i += ngrp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const int64_t bend = sb_ptr[i + 1];
    for (int64_t b = sb_ptr[i] + sub; b < bend; b += NB * LPR) {
      uint2 a0[NB], a1[NB], a2[NB];
      bool live[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {                          // a strip past the end re-reads the first one with its d zeroed
        live[u] = b + u * LPR < bend;
        const int64_t bu = live[u] ? b + u * LPR : b;
        a0[u] = rec[3 * bu]; a1[u] = rec[3 * bu + 1]; a2[u] = rec[3 * bu + 2];
      }
      float4 xv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) xv[u] = d4[a2[u].y];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const float4 v = live[u] ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        s0 += h2f(a0[u].x & 0xffffu) * v.x + h2f(a0[u].x >> 16) * v.y + h2f(a0[u].y & 0xffffu) * v.z;
        s1 += h2f(a0[u].y >> 16) * v.x + h2f(a1[u].x & 0xffffu) * v.y + h2f(a1[u].x >> 16) * v.z;
        s2 += h2f(a1[u].y & 0xffffu) * v.x + h2f(a1[u].y >> 16) * v.y + h2f(a2[u].x & 0xffffu) * v.z;
      }
    }
    s0 = group_sum<LPR>(s0); s1 = group_sum<LPR>(s1); s2 = group_sum<LPR>(s2);
    float rc = 0.f, dc = 0.f;
    if (sub < 3) {
      dc = din[4 * i + sub];
      rc = r[4 * i + sub] - (sub == 0 ? s0 : (sub == 1 ? s1 : s2));
    }
    const float r0 = dpp_f<0x00>(rc), r1 = dpp_f<0x55>(rc), r2 = dpp_f<0xAA>(rc);      // lanes 0-2 of the row, seen from its first quad
    if (sub < 3) {
      const float4 brow = reinterpret_cast<const float4*>(binv12 + 12 * i)[sub];
      x[4 * i + sub] += dc;
      r[4 * i + sub] = rc;
      dout[4 * i + sub] = c1 * dc + c2 * (brow.x * r0 + brow.y * r1 + brow.z * r2);
    }
  }
}
void launch_sweep_sb_h(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const void* rec, const float* binv12, float c1, float c2,
                       const float* din, float* dout, float* x, float* r) {
  constexpr int LPR = 8;
  const int64_t blocks = (nS * LPR + 255) / 256;
  hipLaunchKernelGGL(k_sweep_sb_h<LPR>, dim3(xcd_grid(blocks)), dim3(256), 0, st, nS, sb_ptr, static_cast<const uint2*>(rec), binv12, c1, c2, din, dout, x, r);
}

// The same fused sweep without tiles (coarse level of the displacement block: the vertex graph, L2-resident)
__global__ __launch_bounds__(256) void k_sweep_sc_f32(int64_t N2, const int64_t* __restrict__ nadj_ptr,
                                                      const int32_t* __restrict__ nadj, const float* __restrict__ chat,
                                                      const uint8_t* __restrict__ rowflag, float c1, float c2,
                                                      const float* __restrict__ din, float* __restrict__ dout,
                                                      float* __restrict__ x, float* __restrict__ r) {
  // vertex graph: ~15 entries per row.  4 lanes per row and the four strips of a row issued together: a quarter of the waves
  // of the 16-lane form, and 3 dependent latencies per row (pointer -> index -> gathered d) whatever its length up to 16.
  const int sub = threadIdx.x & 3;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 2;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 2;
  const float4* d4 = reinterpret_cast<const float4*>(din);
  for (int64_t row = grp; row < N2; row += ngrp) {
    const int64_t b = nadj_ptr[row + 1];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t e = nadj_ptr[row] + sub; e < b; e += 16) {
      const bool p1 = e + 4 < b, p2 = e + 8 < b, p3 = e + 12 < b;
      const int64_t e1 = p1 ? e + 4 : e, e2 = p2 ? e + 8 : e, e3 = p3 ? e + 12 : e;
      const int k0 = nadj[e], k1 = nadj[e1], k2 = nadj[e2], k3 = nadj[e3];
      const float a0 = chat[e];
      float a1 = chat[e1], a2 = chat[e2], a3 = chat[e3];
      if (!p1) a1 = 0.f;
      if (!p2) a2 = 0.f;
      if (!p3) a3 = 0.f;
      const float4 x0 = d4[k0], x1 = d4[k1], x2 = d4[k2], x3 = d4[k3];
      s0 += (a0 * x0.x + a1 * x1.x) + (a2 * x2.x + a3 * x3.x);
      s1 += (a0 * x0.y + a1 * x1.y) + (a2 * x2.y + a3 * x3.y);
      s2 += (a0 * x0.z + a1 * x1.z) + (a2 * x2.z + a3 * x3.z);
    }
    s0 = group_sum<4>(s0); s1 = group_sum<4>(s1); s2 = group_sum<4>(s2);
    if (sub == 0) {
      const float4 dr = d4[row], xr = reinterpret_cast<const float4*>(x)[row];
      float4 rr = reinterpret_cast<const float4*>(r)[row];
      if (rowflag[3 * row]) s0 = dr.x;
      if (rowflag[3 * row + 1]) s1 = dr.y;
      if (rowflag[3 * row + 2]) s2 = dr.z;
      rr.x -= s0; rr.y -= s1; rr.z -= s2;
      reinterpret_cast<float4*>(x)[row] = make_float4(xr.x + dr.x, xr.y + dr.y, xr.z + dr.z, 0.f);
      reinterpret_cast<float4*>(r)[row] = make_float4(rr.x, rr.y, rr.z, 0.f);
      reinterpret_cast<float4*>(dout)[row] = make_float4(c1 * dr.x + c2 * rr.x, c1 * dr.y + c2 * rr.y, c1 * dr.z + c2 * rr.z, 0.f);
    }
  }
}
void launch_sweep_sc_f32(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const float* chat,
                         const uint8_t* rowflag, float c1, float c2, const float* din, float* dout, float* x, float* r) {
  int64_t blocks = (N2 + 63) / 64;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_sweep_sc_f32, dim3((unsigned)blocks), dim3(256), 0, st, N2, nadj_ptr, nadj, chat, rowflag, c1, c2, din, dout, x, r);
}
int tile_limit() { return TILE_LIMIT; }

// ---- two-level (P2 -> P1) solve of the displacement block -----------------------------------------------------------
// A_dd = A0 (x) I_3 with A0 the symmetric mass / mesh-Laplacian matrix of the P2 nodes (rows of the assembled matrix are
// equilibrated, so a0_ab = db_ab / rowscale_a).  The P1 functions are a subspace of the P2 functions (vertex values kept,
// edge-midpoint value = mean of the two end vertices), so the Galerkin coarse operator is  A_c = P^T A0 P  on the vertex
// graph.  Chebyshev sweeps on the fine level only have to damp the upper part of the spectrum; the smooth error is
// handled by many, 15x cheaper, sweeps on A_c.
__global__ void k_mg_d0(int64_t N2, const int64_t* __restrict__ nadj_ptr, const int32_t* __restrict__ nadj,
                        const double* __restrict__ db, const double* __restrict__ rowscale,
                        const uint8_t* __restrict__ rowflag, float* __restrict__ d0, int32_t* __restrict__ flags) {
  GS(r, N2) {
    const uint8_t f0 = rowflag[3 * r], f1 = rowflag[3 * r + 1], f2 = rowflag[3 * r + 2];
    if (f0 != f1 || f0 != f2) atomicOr(&flags[1], 32);          // per-component Dirichlet rows: one scalar operator does not fit
    double d = 0.0;
    if (!f0)
      for (int64_t e = nadj_ptr[r]; e < nadj_ptr[r + 1]; ++e) if (nadj[e] == r) d = db[3 * e] / rowscale[6 * r];
    d0[r] = (float)d;
  }
}
// One wave per COARSE row i, lane l owns entry cptr[i] + l of that row and walks, like every other lane of the wave, over
// the row's contributions in one fixed order - children a of i, fine neighbours b of a, parents j of b - keeping what falls
// on its column.  Nothing is scattered: the sums are bitwise reproducible (the fine-row form added with atomics in whatever
// order the waves arrived, and a coarse operator that differs in its last bit moves the eigenvalue estimate, the Chebyshev
// interval and with them every iterate of the outer solver at the level of its tolerance).
__global__ __launch_bounds__(256) void k_mg_rap(int64_t nc, const int64_t* __restrict__ chptr, const int32_t* __restrict__ child,
                                                const float* __restrict__ chw, const int64_t* __restrict__ nadj_ptr,
                                                const int32_t* __restrict__ nadj, const double* __restrict__ db,
                                                const double* __restrict__ rowscale, const uint8_t* __restrict__ rowflag,
                                                const int32_t* __restrict__ par, const float* __restrict__ pw,
                                                const int64_t* __restrict__ cptr, const int32_t* __restrict__ ccol,
                                                double* __restrict__ Ac, int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < nc; i += nwaves) {
    const int64_t c0 = cptr[i], c1 = cptr[i + 1];
    for (int64_t e0 = c0; e0 < c1; e0 += 64) {
      const int64_t e = e0 + lane;
      const int32_t mycol = e < c1 ? ccol[e] : -1;
      double acc = 0.0;
      bool missed = false;
      for (int64_t k = chptr[i]; k < chptr[i + 1]; ++k) {
        const int64_t a = child[k];
        if (rowflag[3 * a]) continue;
        const float wi = chw[k];
        const double inv_sc = 1.0 / rowscale[6 * a];
        for (int64_t ee = nadj_ptr[a]; ee < nadj_ptr[a + 1]; ++ee) {
          const int64_t b = nadj[ee];
          if (rowflag[3 * b]) continue;
          const double aab = db[3 * ee] * inv_sc;
          for (int pj = 0; pj < 2; ++pj) {
            const float wj = pw[2 * b + pj];
            if (wj == 0.f) continue;
            const int32_t j = par[2 * b + pj];
            if (j == mycol) acc += (double)(wi * wj) * aab;
            if (c1 - c0 <= 64 && !__any(j == mycol)) missed = true;     // the vertex graph misses a pair: structure bug
          }
        }
      }
      if (e < c1) Ac[e] = acc;
      if (missed && lane == 0) atomicOr(&flags[1], 64);
    }
  }
}
// Jacobi-scaled coarse operator (one ratio per vertex pair), identity rows where the vertex is a Dirichlet node of the fine
// operator or no free fine node feeds it; dcinv4 = 1 / diag as float4 per coarse node; rowmax = max_i sum_j |cc_ij|
__global__ void k_mg_coarse_finish(int64_t nc, const int64_t* __restrict__ cptr, const int32_t* __restrict__ ccol,
                                   const double* __restrict__ Ac, const int32_t* __restrict__ cfine,
                                   const uint8_t* __restrict__ rowflag, float* __restrict__ cc, uint8_t* __restrict__ cflag,
                                   float* __restrict__ dcinv4, int32_t* __restrict__ rowmax_bits) {
  GS(i, nc) {
    int64_t dg = -1;
    for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) if (ccol[e] == i) dg = e;
    const double d = dg >= 0 ? Ac[dg] : 0.0;
    const bool ident = rowflag[3 * (int64_t)cfine[i]] || !(d > 0.0);
    float sum = 0.f;
    for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) {
      const float v = ident ? (e == dg ? 1.f : 0.f) : (float)(Ac[e] / d);
      cc[e] = v;
      sum += fabsf(v);
    }
    for (int c = 0; c < 3; ++c) { cflag[3 * i + c] = ident ? 1 : 0; dcinv4[4 * i + c] = ident ? 0.f : (float)(1.0 / d); }
    dcinv4[4 * i + 3] = 0.f;
    atomicMax(rowmax_bits, __float_as_int(sum));                   // positive floats order like their bit patterns
  }
}
// coarse right-hand side: D_c^-1 P^T (D0 r), the fine residual r being that of the Jacobi-scaled system
__global__ void k_mg_restrict(int64_t nc, const int64_t* __restrict__ chptr, const int32_t* __restrict__ child,
                              const float* __restrict__ chw, const float* __restrict__ d0, const float* __restrict__ r4,
                              const float* __restrict__ dcinv4, float* __restrict__ rc4, float inv_theta,
                              float* __restrict__ cx, float* __restrict__ cr, float* __restrict__ cd) {
  // four lanes per coarse vertex (a vertex has ~14 children: itself and its edge midpoints): each child is a dependent
  // index -> (weight, residual) gather, and one lane per vertex left 14 of them in a row (96 us per launch at 190 k vertices)
  const int sub = threadIdx.x & 3;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 2;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 2;
  for (int64_t i = grp; i < nc; i += ngrp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t k = chptr[i] + sub; k < chptr[i + 1]; k += 4) {
      const int32_t a = child[k];
      const float w = chw[k] * d0[a];                              // d0 = 0 on Dirichlet rows: they do not feed the coarse level
      const float4 rv = reinterpret_cast<const float4*>(r4)[a];
      s0 += w * rv.x; s1 += w * rv.y; s2 += w * rv.z;
    }
    s0 = group_sum<4>(s0); s1 = group_sum<4>(s1); s2 = group_sum<4>(s2);
    if (sub == 0) {
      const float di = dcinv4[4 * i];
      const float4 rc = make_float4(di * s0, di * s1, di * s2, 0.f);
      reinterpret_cast<float4*>(rc4)[i] = rc;
      if (cx) {                                       // + the coarse level's Chebyshev start (was its own launch): x = 0, r = rhs, d = rhs / theta
        reinterpret_cast<float4*>(cx)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(cr)[i] = rc;
        reinterpret_cast<float4*>(cd)[i] = make_float4(rc.x * inv_theta, rc.y * inv_theta, rc.z * inv_theta, 0.f);
      }
    }
  }
}
__global__ void k_mg_prolong(int64_t N2, const int32_t* __restrict__ par, const float* __restrict__ pw,
                             const float* __restrict__ d0, const float* __restrict__ xc4, float* __restrict__ e4) {
  GS(a, N2) {
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d0[a] != 0.f) {
      const float4 u = reinterpret_cast<const float4*>(xc4)[par[2 * a]], v = reinterpret_cast<const float4*>(xc4)[par[2 * a + 1]];
      const float wu = pw[2 * a], wv = pw[2 * a + 1];
      out = make_float4(wu * u.x + wv * v.x, wu * u.y + wv * v.y, wu * u.z + wv * v.z, 0.f);
    }
    reinterpret_cast<float4*>(e4)[a] = out;
  }
}
void launch_mg_d0(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                  const double* rowscale, const uint8_t* rowflag, float* d0, int32_t* flags) {
  hipLaunchKernelGGL(k_mg_d0, dim3(gridn(N2)), dim3(256), 0, st, N2, nadj_ptr, nadj, db, rowscale, rowflag, d0, flags);
}
void launch_mg_rap(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                   const int64_t* nadj_ptr, const int32_t* nadj, const double* db, const double* rowscale,
                   const uint8_t* rowflag, const int32_t* par, const float* pw, const int64_t* cptr, const int32_t* ccol,
                   double* Ac, int32_t* flags) {
  int64_t blocks = (nc + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_mg_rap, dim3((unsigned)blocks), dim3(256), 0, st, nc, chptr, child, chw, nadj_ptr, nadj, db, rowscale,
                     rowflag, par, pw, cptr, ccol, Ac, flags);
}
void launch_mg_coarse_finish(hipStream_t st, int64_t nc, const int64_t* cptr, const int32_t* ccol, const double* Ac,
                             const int32_t* cfine, const uint8_t* rowflag, float* cc, uint8_t* cflag, float* dcinv4,
                             int32_t* rowmax_bits) {
  hipLaunchKernelGGL(k_mg_coarse_finish, dim3(gridn(nc)), dim3(256), 0, st, nc, cptr, ccol, Ac, cfine, rowflag, cc, cflag,
                     dcinv4, rowmax_bits);
}
void launch_mg_restrict(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                        const float* d0, const float* r4, const float* dcinv4, float* rc4, float inv_theta, float* cx, float* cr,
                        float* cd) {
  hipLaunchKernelGGL(k_mg_restrict, dim3(gridn(4 * nc)), dim3(256), 0, st, nc, chptr, child, chw, d0, r4, dcinv4, rc4, inv_theta, cx, cr, cd);
}
void launch_mg_prolong(hipStream_t st, int64_t N2, const int32_t* par, const float* pw, const float* d0, const float* xc4,
                       float* e4) {
  hipLaunchKernelGGL(k_mg_prolong, dim3(gridn(N2)), dim3(256), 0, st, N2, par, pw, d0, xc4, e4);
}

void launch_extract_chat(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                         float* chat, uint8_t* rowflag, int32_t* flags) {
  hipLaunchKernelGGL(k_extract_chat, dim3(gridn(N2)), dim3(256), 0, st, N2, nadj_ptr, nadj, db, chat, rowflag, flags);
}
void launch_spmv_sc_f32(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const float* chat,
                        const uint8_t* rowflag, const float* x, float* y) {
  int64_t blocks = (N2 + 15) / 16;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_spmv_sc_f32, dim3((unsigned)blocks), dim3(256), 0, st, N2, nadj_ptr, nadj, chat, rowflag, x, y);
}

void launch_spmv_db_f32(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const float* db,
                        const float* x, float* y) {
  int64_t blocks = (N2 + 15) / 16;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_spmv_db_f32, dim3((unsigned)blocks), dim3(256), 0, st, N2, nadj_ptr, nadj, db, x, y);
}
void launch_to_f32(hipStream_t st, int64_t n, const double* a, float* b) { hipLaunchKernelGGL(k_to_f32, dim3(gridn(n)), dim3(256), 0, st, n, a, b); }
void launch_pad_to_f32(hipStream_t st, int64_t nn, const double* a, const float* scale4, float* b) {
  hipLaunchKernelGGL(k_pad_to_f32, dim3(gridn(4 * nn)), dim3(256), 0, st, nn, a, scale4, b);
}
void launch_unpad_from_f32(hipStream_t st, int64_t nn, const float* a, double* b) {
  hipLaunchKernelGGL(k_unpad_from_f32, dim3(gridn(3 * nn)), dim3(256), 0, st, nn, a, b);
}
void launch_dinv_f32(hipStream_t st, int64_t nn, const double* mask, const int64_t* diagpos, const double* A, float* dinv4) {
  hipLaunchKernelGGL(k_dinv_f32, dim3(gridn(4 * nn)), dim3(256), 0, st, nn, mask, diagpos, A, dinv4);
}
void launch_extract_db(hipStream_t st, int64_t N2, int64_t npairs, const int64_t* nadj_ptr, const int64_t* rowptr3,
                       const double* vals, double* db, int32_t* flags, int check) {
  hipLaunchKernelGGL(k_extract_db, dim3(gridn(npairs)), dim3(256), 0, st, N2, nadj_ptr, rowptr3, vals, db, flags, check);
}
void launch_spmv_db(hipStream_t st, int64_t N2, const int64_t* nadj_ptr, const int32_t* nadj, const double* db,
                    const double* x, double* y, const uint8_t* rowmask) {
  int64_t blocks = (N2 + 15) / 16;
  hipLaunchKernelGGL(k_spmv_db, dim3((unsigned)blocks), dim3(256), 0, st, N2, nadj_ptr, nadj, db, rowmask, x, y);
}

// compact solid block: values gathered from Avv~ (vals[e] = src[pos[e]]), vectors gathered / scattered by node list
__global__ void k_gather_vals(int64_t n, const int64_t* __restrict__ pos, const double* __restrict__ src, double* __restrict__ dst) {
  GS(e, n) dst[e] = src[pos[e]];
}
__global__ void k_gather3(int64_t nS, const int32_t* __restrict__ snode, const double* __restrict__ full, double* __restrict__ comp) {
  GS(t, 3 * nS) comp[t] = full[3 * (int64_t)snode[t / 3] + t % 3];
}
__global__ void k_scatter3(int64_t nS, const int32_t* __restrict__ snode, const double* __restrict__ comp, double* __restrict__ full) {
  GS(t, 3 * nS) full[3 * (int64_t)snode[t / 3] + t % 3] = comp[t];
}
void launch_gather_vals(hipStream_t st, int64_t n, const int64_t* pos, const double* src, double* dst) {
  hipLaunchKernelGGL(k_gather_vals, dim3(gridn(n)), dim3(256), 0, st, n, pos, src, dst);
}
void launch_gather3(hipStream_t st, int64_t nS, const int32_t* snode, const double* full, double* comp) {
  hipLaunchKernelGGL(k_gather3, dim3(gridn(3 * nS)), dim3(256), 0, st, nS, snode, full, comp);
}
void launch_scatter3(hipStream_t st, int64_t nS, const int32_t* snode, const double* comp, double* full) {
  hipLaunchKernelGGL(k_scatter3, dim3(gridn(3 * nS)), dim3(256), 0, st, nS, snode, comp, full);
}
// ---- solid velocity block in FP32 block-CSR (3x3 blocks, one column index per block): 40 B per 9 entries ---------------------
// The Chebyshev sweeps on this block deliver ~1e-2 accuracy to a flexible outer method, so single precision is enough;
// it more than halves the bytes of the most often launched kernel.
__global__ void k_sb_gather(int64_t nb, const int32_t* __restrict__ sb_row, const int64_t* __restrict__ sb_src,
                            const int32_t* __restrict__ sb_stride, const double* __restrict__ Avv, float* __restrict__ vals) {
  GS(b, nb) {
    const int64_t src = sb_src[b];
    const int64_t stride = sb_stride[sb_row[b]];
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < 3; ++j) vals[9 * b + 3 * c + j] = (float)Avv[src + c * stride + j];
  }
}
__global__ void k_sb_dinv(int64_t nS, const int32_t* __restrict__ snode, const int64_t* __restrict__ diagpos3,
                          const double* __restrict__ Avv, float* __restrict__ dinv4) {
  GS(t, 4 * nS) dinv4[t] = (t & 3) < 3 ? (float)(1.0 / Avv[diagpos3[3 * (int64_t)snode[t >> 2] + (t & 3)]]) : 0.f;
}
__global__ __launch_bounds__(256) void k_spmv_sb(int64_t nS, const int64_t* __restrict__ sb_ptr,
                                                 const int32_t* __restrict__ sb_col, const float* __restrict__ vals,
                                                 const float* __restrict__ x, float* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t i = grp; i < nS; i += ngrp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t b = sb_ptr[i] + sub; b < sb_ptr[i + 1]; b += 16) {
      const float4 xv = reinterpret_cast<const float4*>(x)[sb_col[b]];      // vectors are float4 per node: one 16-B gather
      const float* a = vals + 9 * b;
      const float x0 = xv.x, x1 = xv.y, x2 = xv.z;
      s0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
      s1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
      s2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    if (sub == 0) reinterpret_cast<float4*>(y)[i] = make_float4(s0, s1, s2, 0.f);
  }
}
__global__ void k_cheb_init_f32(int64_t n, const float* __restrict__ rhs, const float* __restrict__ dinv, float inv_theta,
                                float* __restrict__ x, float* __restrict__ r, float* __restrict__ d) {
  GS(i, n) { const float ri = rhs[i]; x[i] = 0.f; r[i] = ri; d[i] = ri * inv_theta * dinv[i]; }
}
__global__ void k_cheb_step_f32(int64_t n, const float* __restrict__ t, const float* __restrict__ dinv, float c1, float c2,
                                float* __restrict__ x, float* __restrict__ r, float* __restrict__ d) {
  GS(i, n) {
    const float di = d[i], ri = r[i] - t[i];
    x[i] += di;
    r[i] = ri;
    d[i] = c1 * di + c2 * ri * dinv[i];
  }
}
__global__ void k_gather3_f32(int64_t nS, const int32_t* __restrict__ snode, const double* __restrict__ full, float* __restrict__ comp4) {
  GS(t, 4 * nS) comp4[t] = (t & 3) < 3 ? (float)full[3 * (int64_t)snode[t >> 2] + (t & 3)] : 0.f;
}
__global__ void k_scatter3_f32(int64_t nS, const int32_t* __restrict__ snode, const float* __restrict__ comp4, double* __restrict__ full) {
  GS(t, 3 * nS) full[3 * (int64_t)snode[t / 3] + t % 3] = (double)comp4[4 * (t / 3) + t % 3];
}
// ---- 3x3 node-block Jacobi scaling for the solid sweeps: D_b^-1 per solid node (elasticity couples the components
// of a node as strongly as neighbouring nodes; scaling by the block roughly sixths the condition number) ---------------
__global__ void k_sb_binv(int64_t nS, const int32_t* __restrict__ snode, const int64_t* __restrict__ diagpos3,
                          const double* __restrict__ Avv, float* __restrict__ binv12, double* __restrict__ binv9) {
  GS(i, nS) {
    const int64_t r = snode[i];
    double a[3][3];
    for (int c = 0; c < 3; ++c) {
      const double* row = Avv + (diagpos3[3 * r + c] - c);       // start of the diagonal 3x3 block in row c
      for (int j = 0; j < 3; ++j) a[c][j] = row[j];
    }
    const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[0][2] * a[2][1] - a[0][1] * a[2][2],
                 c02 = a[0][1] * a[1][2] - a[0][2] * a[1][1], c10 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                 c11 = a[0][0] * a[2][2] - a[0][2] * a[2][0], c12 = a[0][2] * a[1][0] - a[0][0] * a[1][2],
                 c20 = a[1][0] * a[2][1] - a[1][1] * a[2][0], c21 = a[0][1] * a[2][0] - a[0][0] * a[2][1],
                 c22 = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    const double det = a[0][0] * c00 + a[0][1] * c10 + a[0][2] * c20;
    const double q = 1.0 / det;
    const double inv[3][3] = {{c00 * q, c01 * q, c02 * q}, {c10 * q, c11 * q, c12 * q}, {c20 * q, c21 * q, c22 * q}};
    for (int c = 0; c < 3; ++c) {
      for (int j = 0; j < 3; ++j) { binv9[9 * i + 3 * c + j] = inv[c][j]; binv12[12 * i + 4 * c + j] = (float)inv[c][j]; }
      binv12[12 * i + 4 * c + 3] = 0.f;
    }
  }
}
__global__ void k_block_scale_d(int64_t nS, const double* __restrict__ binv9, double* __restrict__ y) {
  GS(i, nS) {
    const double* b = binv9 + 9 * i;
    const double y0 = y[3 * i], y1 = y[3 * i + 1], y2 = y[3 * i + 2];
    y[3 * i] = b[0] * y0 + b[1] * y1 + b[2] * y2;
    y[3 * i + 1] = b[3] * y0 + b[4] * y1 + b[5] * y2;
    y[3 * i + 2] = b[6] * y0 + b[7] * y1 + b[8] * y2;
  }
}
__device__ inline float4 bmul(const float* __restrict__ b12, float4 v) {
  const float4 r0 = reinterpret_cast<const float4*>(b12)[0], r1 = reinterpret_cast<const float4*>(b12)[1],
               r2 = reinterpret_cast<const float4*>(b12)[2];
  return make_float4(r0.x * v.x + r0.y * v.y + r0.z * v.z, r1.x * v.x + r1.y * v.y + r1.z * v.z,
                     r2.x * v.x + r2.y * v.y + r2.z * v.z, 0.f);
}
__global__ void k_cheb_init_b3(int64_t nS, const float* __restrict__ rhs, const float* __restrict__ binv12, float inv_theta,
                               float* __restrict__ x, float* __restrict__ r, float* __restrict__ d) {
  GS(i, nS) {
    const float4 ri = reinterpret_cast<const float4*>(rhs)[i];
    const float4 z = bmul(binv12 + 12 * i, ri);
    reinterpret_cast<float4*>(x)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(r)[i] = ri;
    reinterpret_cast<float4*>(d)[i] = make_float4(z.x * inv_theta, z.y * inv_theta, z.z * inv_theta, 0.f);
  }
}
// start of the solid two-level cycle in ONE launch (round 5; was gather + memset + init): r = the solid rows of the velocity
// residual, x = 0, d = scale B^-1 r, the second direction buffer zeroed
__global__ void k_solid_cycle_init(int64_t nS, const int32_t* __restrict__ snode, const double* __restrict__ full,
                                   const float* __restrict__ binv12, float scale, float* __restrict__ x, float* __restrict__ r,
                                   float* __restrict__ d, float* __restrict__ d2) {
  GS(i, nS) {
    const int64_t row = 3 * (int64_t)snode[i];
    const float4 ri = make_float4((float)full[row], (float)full[row + 1], (float)full[row + 2], 0.f);
    const float4 z = bmul(binv12 + 12 * i, ri);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(x)[i] = zero;
    reinterpret_cast<float4*>(d2)[i] = zero;
    reinterpret_cast<float4*>(r)[i] = ri;
    reinterpret_cast<float4*>(d)[i] = make_float4(z.x * scale, z.y * scale, z.z * scale, 0.f);
  }
}
__global__ void k_cheb_step_b3(int64_t nS, const float* __restrict__ t, const float* __restrict__ binv12, float c1, float c2,
                               float* __restrict__ x, float* __restrict__ r, float* __restrict__ d) {
  GS(i, nS) {
    const float4 di = reinterpret_cast<float4*>(d)[i], ti = reinterpret_cast<const float4*>(t)[i];
    float4 ri = reinterpret_cast<float4*>(r)[i], xi = reinterpret_cast<float4*>(x)[i];
    ri.x -= ti.x; ri.y -= ti.y; ri.z -= ti.z;
    xi.x += di.x; xi.y += di.y; xi.z += di.z;
    const float4 z = bmul(binv12 + 12 * i, ri);
    reinterpret_cast<float4*>(x)[i] = xi;
    reinterpret_cast<float4*>(r)[i] = ri;
    reinterpret_cast<float4*>(d)[i] = make_float4(c1 * di.x + c2 * z.x, c1 * di.y + c2 * z.y, c1 * di.z + c2 * z.z, 0.f);
  }
}
// ---- two-level (P2 -> P1) solve of the solid velocity block: 3x3-block version of the displacement cycle above ---------
// Fine operator: the FP32 block-CSR copy (rows equilibrated by `rowscale`), undone here so that A_c = P^T A0 P is the
// Galerkin operator of the symmetric elasticity + mass matrix.  Nodes whose rows are identity rows (Dirichlet / ghost)
// take no part in the coarse correction.
__global__ void k_sbmg_flags(int64_t nS, const int64_t* __restrict__ sb_ptr, const int32_t* __restrict__ sb_col,
                             const float* __restrict__ vals, uint8_t* __restrict__ flag) {
  GS(i, nS) {
    bool off[3] = {false, false, false};
    for (int64_t b = sb_ptr[i]; b < sb_ptr[i + 1]; ++b) {
      const bool dg = sb_col[b] == i;
      for (int c = 0; c < 3; ++c)
        for (int j = 0; j < 3; ++j)
          if (!(dg && c == j) && vals[9 * b + 3 * c + j] != 0.f) off[c] = true;
    }
    flag[i] = (off[0] && off[1] && off[2]) ? 0 : 1;
  }
}
// as k_mg_rap: one wave per coarse row, lane = entry (a 3x3 block), contributions gathered in a fixed order, no atomics
__global__ __launch_bounds__(256) void k_sbmg_rap(int64_t nc, const int64_t* __restrict__ chptr, const int32_t* __restrict__ child,
                                                  const float* __restrict__ chw, const int64_t* __restrict__ sb_ptr,
                                                  const int32_t* __restrict__ sb_col, const float* __restrict__ vals,
                                                  const int32_t* __restrict__ snode, const double* __restrict__ rowscale,
                                                  const uint8_t* __restrict__ flag, const int32_t* __restrict__ par,
                                                  const float* __restrict__ pw, const int64_t* __restrict__ cptr,
                                                  const int32_t* __restrict__ ccol, float* __restrict__ cvals,
                                                  int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < nc; i += nwaves) {
    const int64_t c0 = cptr[i], c1 = cptr[i + 1];
    for (int64_t e0 = c0; e0 < c1; e0 += 64) {
      const int64_t e = e0 + lane;
      const int32_t mycol = e < c1 ? ccol[e] : -1;
      float acc[9];
      for (int t = 0; t < 9; ++t) acc[t] = 0.f;
      bool missed = false;
      for (int64_t k = chptr[i]; k < chptr[i + 1]; ++k) {
        const int64_t a = child[k];
        if (flag[a]) continue;
        const float wi = chw[k];
        const int64_t r = snode[a];
        const float is0 = (float)(1.0 / rowscale[6 * r + 3]), is1 = (float)(1.0 / rowscale[6 * r + 4]), is2 = (float)(1.0 / rowscale[6 * r + 5]);
        for (int64_t ee = sb_ptr[a]; ee < sb_ptr[a + 1]; ++ee) {
          const int64_t b = sb_col[ee];
          if (flag[b]) continue;
          for (int pj = 0; pj < 2; ++pj) {
            const float wj = pw[2 * b + pj];
            if (wj == 0.f) continue;
            const int32_t j = par[2 * b + pj];
            if (j == mycol) {
              const float ww = wi * wj;
              for (int t = 0; t < 3; ++t) {
                acc[t] += ww * (vals[9 * ee + t] * is0);
                acc[3 + t] += ww * (vals[9 * ee + 3 + t] * is1);
                acc[6 + t] += ww * (vals[9 * ee + 6 + t] * is2);
              }
            }
            if (c1 - c0 <= 64 && !__any(j == mycol)) missed = true;
          }
        }
      }
      if (e < c1) for (int t = 0; t < 9; ++t) cvals[9 * e + t] = acc[t];
      if (missed && lane == 0) atomicOr(&flags[1], 64);
    }
  }
}
// coarse rows: inverse of the diagonal block (block-Jacobi scaling), identity rows for flagged / singular vertices, and a
// bound for the largest eigenvalue of B^-1 A_c (largest absolute row sum of the scaled blocks)
__global__ void k_sbmg_coarse_finish(int64_t nc, const int64_t* __restrict__ cptr, const int32_t* __restrict__ ccol,
                                     float* __restrict__ cvals, const int32_t* __restrict__ cfine,
                                     const uint8_t* __restrict__ flag, float* __restrict__ cbinv12, uint8_t* __restrict__ cflag,
                                     int32_t* __restrict__ rowmax_bits) {
  GS(i, nc) {
    int64_t dg = -1;
    for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) if (ccol[e] == i) dg = e;
    float a[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (dg >= 0) for (int c = 0; c < 3; ++c) for (int j = 0; j < 3; ++j) a[c][j] = cvals[9 * dg + 3 * c + j];
    const float c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[0][2] * a[2][1] - a[0][1] * a[2][2],
                c02 = a[0][1] * a[1][2] - a[0][2] * a[1][1], c10 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                c11 = a[0][0] * a[2][2] - a[0][2] * a[2][0], c12 = a[0][2] * a[1][0] - a[0][0] * a[1][2],
                c20 = a[1][0] * a[2][1] - a[1][1] * a[2][0], c21 = a[0][1] * a[2][0] - a[0][0] * a[2][1],
                c22 = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    const float det = a[0][0] * c00 + a[0][1] * c10 + a[0][2] * c20;
    const bool ident = flag[cfine[i]] || dg < 0 || !(det > 0.f) || !(a[0][0] > 0.f);
    float inv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (!ident) {
      const float q = 1.f / det;
      const float t[3][3] = {{c00 * q, c01 * q, c02 * q}, {c10 * q, c11 * q, c12 * q}, {c20 * q, c21 * q, c22 * q}};
      for (int c = 0; c < 3; ++c) for (int j = 0; j < 3; ++j) inv[c][j] = t[c][j];
    }
    float rs[3] = {0.f, 0.f, 0.f};
    for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) {
      if (ident) { for (int t = 0; t < 9; ++t) cvals[9 * e + t] = (e == dg && (t == 0 || t == 4 || t == 8)) ? 1.f : 0.f; continue; }
      for (int c = 0; c < 3; ++c)
        for (int j = 0; j < 3; ++j)
          rs[c] += fabsf(inv[c][0] * cvals[9 * e + j] + inv[c][1] * cvals[9 * e + 3 + j] + inv[c][2] * cvals[9 * e + 6 + j]);
    }
    for (int c = 0; c < 3; ++c) {
      for (int j = 0; j < 3; ++j) cbinv12[12 * i + 4 * c + j] = inv[c][j];
      cbinv12[12 * i + 4 * c + 3] = 0.f;
    }
    cflag[i] = ident ? 1 : 0;
    atomicMax(rowmax_bits, __float_as_int(ident ? 1.f : fmaxf(rs[0], fmaxf(rs[1], rs[2]))));
  }
}
// coarse rhs = P^T (r / rowscale) of the free fine nodes (zero on identity vertices)
__global__ void k_sbmg_restrict(int64_t nc, const int64_t* __restrict__ chptr, const int32_t* __restrict__ child,
                                const float* __restrict__ chw, const int32_t* __restrict__ snode,
                                const double* __restrict__ rowscale, const uint8_t* __restrict__ flag,
                                const uint8_t* __restrict__ cflag, const float* __restrict__ r4, float* __restrict__ rc4,
                                const int32_t* __restrict__ bpos, double* __restrict__ bd) {
  // 8 lanes per coarse node (round 4; one thread walking its ~20 children through three dependent loads each took 50 us per
  // application at 1.12 M tets): the children's terms are added in a fixed tree order, so the sum stays reproducible
  const int sub = threadIdx.x & 7;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 3;
  for (int64_t i = grp; i < nc; i += ngrp) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (!cflag[i])
      for (int64_t k = chptr[i] + sub; k < chptr[i + 1]; k += 8) {
        const int32_t a = child[k];
        if (flag[a]) continue;
        const int64_t r = snode[a];
        const float w = chw[k];
        const float4 rv = reinterpret_cast<const float4*>(r4)[a];
        s0 += w * rv.x / (float)rowscale[6 * r + 3]; s1 += w * rv.y / (float)rowscale[6 * r + 4]; s2 += w * rv.z / (float)rowscale[6 * r + 5];
      }
    s0 = group_sum<8>(s0); s1 = group_sum<8>(s1); s2 = group_sum<8>(s2);
    if (sub == 0) {
      reinterpret_cast<float4*>(rc4)[i] = make_float4(s0, s1, s2, 0.f);
      if (bd) {                                     // exact coarse solve (fsi_bcr.hip): the right-hand side in its own order, FP64
        const int64_t p = 3 * (int64_t)bpos[i];
        bd[p] = s0; bd[p + 1] = s1; bd[p + 2] = s2;
      }
    }
  }
}
__global__ void k_sbmg_prolong(int64_t nS, const int32_t* __restrict__ par, const float* __restrict__ pw,
                               const uint8_t* __restrict__ flag, const float* __restrict__ xc4, float* __restrict__ e4,
                               const int32_t* __restrict__ bpos, const double* __restrict__ xd) {
  GS(a, nS) {
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!flag[a]) {
      float4 u, v;
      if (xd) {                                     // the exact coarse solve's answer, read where it lies (FP64, its own order)
        const int64_t pu = 3 * (int64_t)bpos[par[2 * a]], pv = 3 * (int64_t)bpos[par[2 * a + 1]];
        u = make_float4((float)xd[pu], (float)xd[pu + 1], (float)xd[pu + 2], 0.f);
        v = make_float4((float)xd[pv], (float)xd[pv + 1], (float)xd[pv + 2], 0.f);
      } else {
        u = reinterpret_cast<const float4*>(xc4)[par[2 * a]];
        v = reinterpret_cast<const float4*>(xc4)[par[2 * a + 1]];
      }
      const float wu = pw[2 * a], wv = pw[2 * a + 1];
      out = make_float4(wu * u.x + wv * v.x, wu * u.y + wv * v.y, wu * u.z + wv * v.z, 0.f);
    }
    reinterpret_cast<float4*>(e4)[a] = out;
  }
}
void launch_sbmg_flags(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const int32_t* sb_col, const float* vals, uint8_t* flag) {
  hipLaunchKernelGGL(k_sbmg_flags, dim3(gridn(nS)), dim3(256), 0, st, nS, sb_ptr, sb_col, vals, flag);
}
void launch_sbmg_rap(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                     const int64_t* sb_ptr, const int32_t* sb_col, const float* vals, const int32_t* snode,
                     const double* rowscale, const uint8_t* flag, const int32_t* par, const float* pw, const int64_t* cptr,
                     const int32_t* ccol, float* cvals, int32_t* flags) {
  int64_t blocks = (nc + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_sbmg_rap, dim3((unsigned)blocks), dim3(256), 0, st, nc, chptr, child, chw, sb_ptr, sb_col, vals, snode,
                     rowscale, flag, par, pw, cptr, ccol, cvals, flags);
}
void launch_sbmg_coarse_finish(hipStream_t st, int64_t nc, const int64_t* cptr, const int32_t* ccol, float* cvals,
                               const int32_t* cfine, const uint8_t* flag, float* cbinv12, uint8_t* cflag, int32_t* rowmax_bits) {
  hipLaunchKernelGGL(k_sbmg_coarse_finish, dim3(gridn(nc)), dim3(256), 0, st, nc, cptr, ccol, cvals, cfine, flag, cbinv12, cflag,
                     rowmax_bits);
}
void launch_sbmg_restrict(hipStream_t st, int64_t nc, const int64_t* chptr, const int32_t* child, const float* chw,
                          const int32_t* snode, const double* rowscale, const uint8_t* flag, const uint8_t* cflag,
                          const float* r4, float* rc4, const int32_t* bpos, double* bd) {
  hipLaunchKernelGGL(k_sbmg_restrict, dim3(gridn(8 * nc)), dim3(256), 0, st, nc, chptr, child, chw, snode, rowscale, flag, cflag, r4, rc4, bpos, bd);
}
void launch_sbmg_prolong(hipStream_t st, int64_t nS, const int32_t* par, const float* pw, const uint8_t* flag, const float* xc4,
                         float* e4, const int32_t* bpos, const double* xd) {
  hipLaunchKernelGGL(k_sbmg_prolong, dim3(gridn(nS)), dim3(256), 0, st, nS, par, pw, flag, xc4, e4, bpos, xd);
}

// The same sweep with only the matrix VALUES in FP32 and every vector in FP64.  A rounded matrix is still one fixed linear
// operator, so the preconditioner stays a linear map (what the Krylov method assumes); FP32 vectors instead add 6e-8 |dp| of
// noise per sweep, which on a coarse mesh with a pressure-dominated right-hand side exceeded the velocity residual the outer
// iteration was trying to reduce (tests/test_gpu_parity.py::test_properties_on_generated_mesh made no progress at all).
// The vectors are 5 V doubles against ~65 V matrix entries: the bytes per sweep are those of the all-FP32 form.
template <int LPR, int KS, class VT = float>       // lanes per row, strips issued together: LPR x KS entries per round; VT: value storage
__global__ __launch_bounds__(256) void k_sweep_csr_mixed(int64_t n, const int64_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ cols, const VT* __restrict__ vals,
                                                         const int64_t* __restrict__ diagpos, const double* __restrict__ dvals,
                                                         double c1, double c2, const double* __restrict__ din,
                                                         double* __restrict__ dout, double* __restrict__ x, double* __restrict__ r) {
  // The sweep is a chain of dependent loads (row pointer -> column index -> gathered d) per row, so all strips of a ~65-entry
  // row are issued together (3 latencies per row whatever its length up to LPR x KS), and a workgroup's rows are consecutive:
  // their sums go through LDS to the first lanes of the workgroup, which do the Chebyshev update on consecutive entries
  // (one lane per row doing it costs 8 memory instructions with 1 lane in LPR active).  Strips past the end re-read the
  // first strip (cache hit) with weight zero.
  constexpr int RPB = 256 / LPR;                 // rows per workgroup and round
  __shared__ double ssum[RPB];
  const int sub = threadIdx.x & (LPR - 1), g = threadIdx.x / LPR;
  for (int64_t base = (int64_t)blockIdx.x * RPB; base < n; base += (int64_t)gridDim.x * RPB) {
    const int64_t i = base + g;
    double s = 0.0;
    if (i < n) {
      const int64_t b = rowptr[i + 1];
      for (int64_t e = rowptr[i] + sub; e < b; e += LPR * KS) {
        int k[KS];
        VT v[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const bool in = e + j * LPR < b;
          const int64_t ej = in ? e + j * LPR : e;
          k[j] = cols[ej];
          v[j] = vals[ej];
          if (!in) v[j] = (VT)0;
        }
        double dj[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) dj[j] = din[k[j]];
#pragma unroll
        for (int j = 0; j < KS; ++j) s += (double)v[j] * dj[j];
      }
    }
s = group_sum<LPR>(s);
    if (sub == 0) ssum[g] = s;
    __syncthreads();
    if (threadIdx.x < RPB && base + threadIdx.x < n) {
      const int64_t row = base + threadIdx.x;
      const double di = din[row], ri = r[row] - ssum[threadIdx.x];
      x[row] += di;
      r[row] = ri;
      dout[row] = c1 * di + c2 * ri / dvals[diagpos[row]];
    }
    __syncthreads();
  }
}
void launch_sweep_csr_mixed(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const float* vals,
                            const int64_t* diagpos, const double* dvals, double c1, double c2, const double* din, double* dout,
                            double* x, double* r) {
  constexpr int lpr = 8;            // lanes per row of the generic Schur sweep (round 2 scan: 8)
  const int rpb = 256 / (lpr == 16 ? 16 : (lpr == 4 ? 4 : 8));
  int64_t blocks = (n + rpb - 1) / rpb;
  if (blocks > 32768) blocks = 32768;
  if (lpr == 16)
    hipLaunchKernelGGL((k_sweep_csr_mixed<16, 4>), dim3((unsigned)blocks), dim3(256), 0, st, n, rowptr, cols, vals, diagpos, dvals, c1, c2, din, dout, x, r);
  else if (lpr == 4)
    hipLaunchKernelGGL((k_sweep_csr_mixed<4, 16>), dim3((unsigned)blocks), dim3(256), 0, st, n, rowptr, cols, vals, diagpos, dvals, c1, c2, din, dout, x, r);
  else
    hipLaunchKernelGGL((k_sweep_csr_mixed<8, 8>), dim3((unsigned)blocks), dim3(256), 0, st, n, rowptr, cols, vals, diagpos, dvals, c1, c2, din, dout, x, r);
}
// all-FP64 storage (FSI_SCHUR_FP32=0, the "FP64" reading of BASELINE configs[1]): the same fused sweep on the FP64 values;
// round 2 ran this case as a generic CSR product plus a separate update kernel (two launches, 54 us per sweep)
void launch_sweep_csr_f64(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const double* vals,
                          const int64_t* diagpos, double c1, double c2, const double* din, double* dout, double* x, double* r) {
  int64_t blocks = (n + 31) / 32;
  if (blocks > 32768) blocks = 32768;
  hipLaunchKernelGGL((k_sweep_csr_mixed<8, 8, double>), dim3((unsigned)blocks), dim3(256), 0, st, n, rowptr, cols, vals, diagpos, vals, c1, c2,
                     din, dout, x, r);
}
// The Schur sweep on packed records (FP16 value + 16-bit tile-local column, 4 bytes per entry instead of 8) with the d entries
// of a 256-row tile's columns staged once in LDS (FP64; the tile of a two-ring pattern sees ~2-3 k distinct columns).
// Vectors stay FP64 as in k_sweep_csr_mixed; a rounded matrix is still one linear operator.
// SCHUR_TILE rows per workgroup (64 | 128 | 256, chosen per context from the number of pressure rows: FsiTuning.schur_tile_rows).
// A sweep is a chain of dependent steps per workgroup - stage the tile's distinct columns, 32 rows per pass, update - and at the
// per-GPU sizes of a partitioned run (24 k rows: 94 tiles of 256 rows on 256 CUs) its length, not the chip, sets the time.
template <int SCHUR_TILE>
__global__ __launch_bounds__(256) void k_sweep_schur_tiled(int64_t n, const int64_t* __restrict__ rowptr,
                                                           const uint32_t* __restrict__ rec, const int64_t* __restrict__ tile_uptr,
                                                           const int32_t* __restrict__ ulist, const double* __restrict__ dinv,
                                                           double c1, double c2, const double* __restrict__ din,
                                                           double* __restrict__ dout, double* __restrict__ x, double* __restrict__ r) {
  extern __shared__ __attribute__((aligned(16))) double sxd[];
  __shared__ int64_t sptr[SCHUR_TILE + 1];
  __shared__ double ssum[SCHUR_TILE];
  const int64_t tile = xcd_tile((n + SCHUR_TILE - 1) / SCHUR_TILE);
  if (tile < 0) return;
  const int64_t r0 = tile * SCHUR_TILE;
  const int nrows = (int)((r0 + SCHUR_TILE < n ? r0 + SCHUR_TILE : n) - r0);
  const int64_t u0 = tile_uptr[tile], nu = tile_uptr[tile + 1] - u0;
  for (int64_t i = threadIdx.x; i < nu; i += 1024) {              // four dependent index -> entry pairs in flight
    const int64_t i1 = i + 256, i2 = i + 512, i3 = i + 768;
    const int32_t k0 = ulist[u0 + i], k1 = i1 < nu ? ulist[u0 + i1] : 0, k2 = i2 < nu ? ulist[u0 + i2] : 0, k3 = i3 < nu ? ulist[u0 + i3] : 0;
    const double v0 = din[k0], v1 = din[k1], v2 = din[k2], v3 = din[k3];
    sxd[i] = v0;
    if (i1 < nu) sxd[i1] = v1;
    if (i2 < nu) sxd[i2] = v2;
    if (i3 < nu) sxd[i3] = v3;
  }
  for (int i = threadIdx.x; i <= nrows; i += 256) sptr[i] = rowptr[r0 + i];
  __syncthreads();
  const int sub = threadIdx.x & 7, g = threadIdx.x >> 3;          // 8 lanes per row, 32 rows per pass
  for (int row = g; row < nrows; row += 32) {
    const int64_t b = sptr[row + 1];
    double s = 0.0;
    for (int64_t e = sptr[row] + sub; e < b; e += 64) {
      uint32_t q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = e + 8 * j < b ? rec[e + 8 * j] : 0u;      // zero record: value +0, local index 0
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (double)h2f(q[j] & 0xffffu) * sxd[q[j] >> 16];
    }
    s = group_sum<8>(s);
    if (sub == 0) ssum[row] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < nrows) {
    const int64_t row = r0 + threadIdx.x;
    const double di = din[row], ri = r[row] - ssum[threadIdx.x];
    x[row] += di;
    r[row] = ri;
    dout[row] = c1 * di + c2 * ri * dinv[row];
  }
}
void launch_sweep_schur_tiled(hipStream_t st, int tile_rows, int64_t n, int max_nu, const int64_t* rowptr, const uint32_t* rec,
                              const int64_t* tile_uptr, const int32_t* ulist, const double* dinv, double c1, double c2,
                              const double* din, double* dout, double* x, double* r) {
  const unsigned tiles = xcd_grid((n + tile_rows - 1) / tile_rows);
  const size_t lds = (size_t)max_nu * sizeof(double);
  if (tile_rows == 32)
    hipLaunchKernelGGL(k_sweep_schur_tiled<32>, dim3(tiles), dim3(256), lds, st, n, rowptr, rec, tile_uptr, ulist, dinv, c1, c2, din, dout, x, r);
  else if (tile_rows == 64)
    hipLaunchKernelGGL(k_sweep_schur_tiled<64>, dim3(tiles), dim3(256), lds, st, n, rowptr, rec, tile_uptr, ulist, dinv, c1, c2, din, dout, x, r);
  else if (tile_rows == 128)
    hipLaunchKernelGGL(k_sweep_schur_tiled<128>, dim3(tiles), dim3(256), lds, st, n, rowptr, rec, tile_uptr, ulist, dinv, c1, c2, din, dout, x, r);
  else
    hipLaunchKernelGGL(k_sweep_schur_tiled<256>, dim3(tiles), dim3(256), lds, st, n, rowptr, rec, tile_uptr, ulist, dinv, c1, c2, din, dout, x, r);
}
// One Chebyshev sweep of the solid block in a single launch: t = A d_in (3x3 block-CSR, 16 lanes per node), then on the
// first three lanes of the group (one component each)  r -= t,  x += d_in,  d_out = c1 d_in + c2 B^-1 r.
// d is ping-ponged because other nodes still gather d_in; the product never goes through memory.
template <int LEVEL>      // LEVEL only names the instantiation: 0 = solid nodes, 1 = coarse level (solid vertices) of the two-level cycle
__global__ __launch_bounds__(256) void k_sweep_sb_b3(int64_t nS, const int64_t* __restrict__ sb_ptr,
                                                     const int32_t* __restrict__ sb_col, const float* __restrict__ vals,
                                                     const float* __restrict__ binv12, float c1, float c2,
                                                     const float* __restrict__ din, float* __restrict__ dout,
                                                     float* __restrict__ x, float* __restrict__ r) {
  const int sub = threadIdx.x & 15;
  // (workgroups in the XCD-aware order of xcd_tile; a grid-stride loop over the logical workgroups keeps the launch's cap)
  const int64_t nlb = (nS + 15) / 16, nwg = gridDim.x, span = xcd_span(nlb);
  for (int64_t lb0 = blockIdx.x; lb0 < span; lb0 += nwg) {
  const int64_t lb = xcd_unit(lb0, nlb);
  const int64_t i = lb * 16 + (threadIdx.x >> 4);
  if (lb >= 0 && i < nS) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const int64_t bend = sb_ptr[i + 1];
    if (LEVEL == 0) {
      // two 16-block strips per round, issued together (index -> gathered d is a dependent pair of loads; an edge node's ~22
      // blocks take one round instead of two, a vertex node's ~65 three instead of five); the strip past the end re-reads
      // the first one with its d zeroed
      for (int64_t b = sb_ptr[i] + sub; b < bend; b += 32) {
        const bool p1 = b + 16 < bend;
        const int64_t b1 = p1 ? b + 16 : b;
        const int k0 = sb_col[b], k1 = sb_col[b1];
        const float* a = vals + 9 * b;
        const float* q = vals + 9 * b1;
        const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8];
        const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8];
        const float4 xv = reinterpret_cast<const float4*>(din)[k0];
        float4 yv = reinterpret_cast<const float4*>(din)[k1];
        if (!p1) yv = make_float4(0.f, 0.f, 0.f, 0.f);
        s0 += (a0 * xv.x + a1 * xv.y + a2 * xv.z) + (q0 * yv.x + q1 * yv.y + q2 * yv.z);
        s1 += (a3 * xv.x + a4 * xv.y + a5 * xv.z) + (q3 * yv.x + q4 * yv.y + q5 * yv.z);
        s2 += (a6 * xv.x + a7 * xv.y + a8 * xv.z) + (q6 * yv.x + q7 * yv.y + q8 * yv.z);
      }
    } else {
      for (int64_t b = sb_ptr[i] + sub; b < bend; b += 16) {
        const float4 xv = reinterpret_cast<const float4*>(din)[sb_col[b]];
        const float* a = vals + 9 * b;
        const float x0 = xv.x, x1 = xv.y, x2 = xv.z;
        s0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
        s1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
        s2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
      }
    }
    s0 = group_sum<16>(s0); s1 = group_sum<16>(s1); s2 = group_sum<16>(s2);
    float rc = 0.f, dc = 0.f;
    if (sub < 3) {
      dc = din[4 * i + sub];
      rc = r[4 * i + sub] - (sub == 0 ? s0 : (sub == 1 ? s1 : s2));
    }
    const float r0 = dpp_f<0x00>(rc), r1 = dpp_f<0x55>(rc), r2 = dpp_f<0xAA>(rc);      // lanes 0-2 of the row, seen from its first quad
    if (sub < 3) {
      const float4 brow = reinterpret_cast<const float4*>(binv12 + 12 * i)[sub];
      x[4 * i + sub] += dc;
      r[4 * i + sub] = rc;
      dout[4 * i + sub] = c1 * dc + c2 * (brow.x * r0 + brow.y * r1 + brow.z * r2);
    }
  }
  }
}
void launch_sweep_sb_b3(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const int32_t* sb_col, const float* vals,
                        const float* binv12, float c1, float c2, const float* din, float* dout, float* x, float* r, int level) {
  int64_t blocks = xcd_grid((nS + 15) / 16);
  if (blocks > 16384) blocks = 16384;                   // (a multiple of 8: the XCD of a logical workgroup is that of its launch index)
  if (level == 0)
    hipLaunchKernelGGL(k_sweep_sb_b3<0>, dim3((unsigned)blocks), dim3(256), 0, st, nS, sb_ptr, sb_col, vals, binv12, c1, c2, din,
                       dout, x, r);
  else
    hipLaunchKernelGGL(k_sweep_sb_b3<1>, dim3((unsigned)blocks), dim3(256), 0, st, nS, sb_ptr, sb_col, vals, binv12, c1, c2, din,
                       dout, x, r);
}
void launch_sb_binv(hipStream_t st, int64_t nS, const int32_t* snode, const int64_t* diagpos3, const double* Avv,
                    float* binv12, double* binv9) {
  hipLaunchKernelGGL(k_sb_binv, dim3(gridn(nS)), dim3(256), 0, st, nS, snode, diagpos3, Avv, binv12, binv9);
}
void launch_block_scale_d(hipStream_t st, int64_t nS, const double* binv9, double* y) {
  hipLaunchKernelGGL(k_block_scale_d, dim3(gridn(nS)), dim3(256), 0, st, nS, binv9, y);
}
void launch_cheb_init_b3(hipStream_t st, int64_t nS, const float* rhs, const float* binv12, float inv_theta, float* x, float* r, float* d) {
  hipLaunchKernelGGL(k_cheb_init_b3, dim3(gridn(nS)), dim3(256), 0, st, nS, rhs, binv12, inv_theta, x, r, d);
}
void launch_cheb_step_b3(hipStream_t st, int64_t nS, const float* t, const float* binv12, float c1, float c2, float* x, float* r, float* d) {
  hipLaunchKernelGGL(k_cheb_step_b3, dim3(gridn(nS)), dim3(256), 0, st, nS, t, binv12, c1, c2, x, r, d);
}
void launch_sb_gather(hipStream_t st, int64_t nb, const int32_t* sb_row, const int64_t* sb_src, const int32_t* sb_stride,
                      const double* Avv, float* vals) {
  hipLaunchKernelGGL(k_sb_gather, dim3(gridn(nb)), dim3(256), 0, st, nb, sb_row, sb_src, sb_stride, Avv, vals);
}
void launch_sb_dinv(hipStream_t st, int64_t nS, const int32_t* snode, const int64_t* diagpos3, const double* Avv, float* dinv) {
  hipLaunchKernelGGL(k_sb_dinv, dim3(gridn(4 * nS)), dim3(256), 0, st, nS, snode, diagpos3, Avv, dinv);
}
void launch_spmv_sb(hipStream_t st, int64_t nS, const int64_t* sb_ptr, const int32_t* sb_col, const float* vals,
                    const float* x, float* y) {
  int64_t blocks = (nS + 15) / 16;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_spmv_sb, dim3((unsigned)blocks), dim3(256), 0, st, nS, sb_ptr, sb_col, vals, x, y);
}
void launch_cheb_init_f32(hipStream_t st, int64_t n, const float* rhs, const float* dinv, float inv_theta, float* x, float* r, float* d) {
  hipLaunchKernelGGL(k_cheb_init_f32, dim3(gridn(n)), dim3(256), 0, st, n, rhs, dinv, inv_theta, x, r, d);
}
void launch_cheb_step_f32(hipStream_t st, int64_t n, const float* t, const float* dinv, float c1, float c2, float* x, float* r, float* d) {
  hipLaunchKernelGGL(k_cheb_step_f32, dim3(gridn(n)), dim3(256), 0, st, n, t, dinv, c1, c2, x, r, d);
}
void launch_gather3_f32(hipStream_t st, int64_t nS, const int32_t* snode, const double* full, float* comp) {
  hipLaunchKernelGGL(k_gather3_f32, dim3(gridn(4 * nS)), dim3(256), 0, st, nS, snode, full, comp);
}
void launch_scatter3_f32(hipStream_t st, int64_t nS, const int32_t* snode, const float* comp, double* full) {
  hipLaunchKernelGGL(k_scatter3_f32, dim3(gridn(3 * nS)), dim3(256), 0, st, nS, snode, comp, full);
}
void launch_solid_cycle_init(hipStream_t st, int64_t nS, const int32_t* snode, const double* full, const float* binv12, float scale,
                             float* x, float* r, float* d, float* d2) {
  hipLaunchKernelGGL(k_solid_cycle_init, dim3(gridn(nS)), dim3(256), 0, st, nS, snode, full, binv12, scale, x, r, d, d2);
}

// x = mask .* (pseudo-random +-1 ripple): start vector of the power iteration, rich in element-scale modes
__global__ void k_mask_ripple(int64_t n, const double* __restrict__ mask, double* __restrict__ x) {
  GS(i, n) {
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    x[i] = (mask ? mask[i] : 1.0) * (((double)(h & 0xFFFF) / 65535.0) - 0.5);
  }
}
void launch_mask_ripple(hipStream_t st, int64_t n, const double* mask, double* x) {
  hipLaunchKernelGGL(k_mask_ripple, dim3(gridn(n)), dim3(256), 0, st, n, mask, x);
}
void launch_mask_scale(hipStream_t st, int64_t n, const double* mask, const int64_t* diagpos, const double* A, double* y) {
  hipLaunchKernelGGL(k_mask_scale, dim3(gridn(n)), dim3(256), 0, st, n, mask, diagpos, A, y);
}

// y = b - A x, one wave per row (generic CSR)
__global__ __launch_bounds__(256) void k_residual_csr(int64_t n, const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                                      const double* __restrict__ x, const double* __restrict__ b,
                                                      double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wave; row < n; row += nwaves) {
    double s = 0.0;
    for (int64_t t = rowptr[row] + lane; t < rowptr[row + 1]; t += 64) s += vals[t] * x[cols[t]];
    s = wsum(s);
    if (lane == 0) y[row] = b[row] - s;
  }
}
// y[rows] = b[rows] - sum_t vals[src[t]] x[col[t]] on a short list of rows: the fluid rows next to the wall, whose only
// coupling to the solid predictor is through a few solid columns (y = b elsewhere, set by the caller).  16 lanes per row.
__global__ __launch_bounds__(256) void k_residual_rows(int64_t nrows, const int32_t* __restrict__ rows,
                                                       const int64_t* __restrict__ ptr, const int32_t* __restrict__ col,
                                                       const int64_t* __restrict__ src, const double* __restrict__ vals,
                                                       const double* __restrict__ x, const double* __restrict__ b,
                                                       double* __restrict__ y) {
  const int sub = threadIdx.x & 15;
  const int64_t grp = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t i = grp; i < nrows; i += ngrp) {
    double s = 0.0;
    for (int64_t t = ptr[i] + sub; t < ptr[i + 1]; t += 16) s += vals[src[t]] * x[col[t]];
    s = group_sum<16>(s);
    if (sub == 0) { const int32_t r = rows[i]; y[r] = b[r] - s; }
  }
}
void launch_residual_rows(hipStream_t st, int64_t nrows, const int32_t* rows, const int64_t* ptr, const int32_t* col,
                          const int64_t* src, const double* vals, const double* x, const double* b, double* y) {
  if (nrows <= 0) return;
  int64_t blocks = (nrows + 15) / 16;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_residual_rows, dim3((unsigned)blocks), dim3(256), 0, st, nrows, rows, ptr, col, src, vals, x, b, y);
}
void launch_residual_csr(hipStream_t st, int64_t n, const int64_t* rowptr, const int32_t* cols, const double* vals,
                         const double* x, const double* b, double* y) {
  int64_t blocks = (n + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_residual_csr, dim3((unsigned)blocks), dim3(256), 0, st, n, rowptr, cols, vals, x, b, y);
}

}  // namespace fsi

// ---- helpers of the coarse levels' power iteration (largest eigenvalue of the scaled operator, in the sweeps' own layout) --
namespace fsi {
__global__ void k_f32_ripple4(int64_t n, float* __restrict__ x) {          // float4 per node, pad lane zero
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
    const float a = ((h & 1023) / 512.0f - 1.0f), b = (((h >> 10) & 1023) / 512.0f - 1.0f), c = (((h >> 20) & 1023) / 512.0f - 1.0f);
    reinterpret_cast<float4*>(x)[i] = make_float4(a, b, c, 0.f);
  }
}
// one workgroup, fixed order of the partial sums: the eigenvalue estimate (and with it the Chebyshev interval of a coarse
// level) is the same in every run; the vectors are the coarse levels' (a few 1e5 entries)
__global__ __launch_bounds__(1024) void k_f32_sumsq(int64_t n, const float* __restrict__ x, double* __restrict__ out) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)x[i] * x[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}
void launch_f32_ripple4(hipStream_t st, int64_t nnodes, float* x) {
  hipLaunchKernelGGL(k_f32_ripple4, dim3(gridn(nnodes)), dim3(256), 0, st, nnodes, x);
}
void launch_f32_sumsq(hipStream_t st, int64_t n, const float* x, double* out) {
  hipLaunchKernelGGL(k_f32_sumsq, dim3(1), dim3(1024), 0, st, n, x, out);
}
}  // namespace fsi
