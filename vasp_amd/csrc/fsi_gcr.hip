// Orthogonalisation kernels of the recycled GCR (solve_gcr in fsi_capi.hip).
//
// The kept space is  Q = A P  (orthonormal columns) and the search directions P.  What streams through HBM in every
// Krylov iteration is Q only: the coefficients h = Q^T w (k_gcr_dots) and the update w -= Q h (k_gcr_axpy, which also
// returns |w'|^2 and w'.r so that the iteration needs two host reads instead of five).  P is touched once per solve
// (k_gcr_flush): the directions made during a solve are kept as the raw preconditioned vectors z plus a small
// coefficient matrix on the host, x is accumulated as coefficients, and one pass over the store turns both into vectors.
//
// Q may be stored in FP32 (template parameter QT): it halves the dominant stream.  All sums, w, r, z and P stay FP64;
// the rounding of q (6e-8) bounds how far one solve cycle can reduce the residual, and solve_gcr restarts a cycle from
// the true residual when a tighter tolerance is asked for.
//
// Layout: column k of Q starts at Q + k * ldq (ldq = n rounded up to 4, so every column is 16-byte aligned for float4 /
// double2 loads); column k of Z at Z + k * ldz (ldz = n rounded up to 2).  Algorithmic bytes per launch:
//   k_gcr_dots   m ldq sizeof(QT) + n 8 (+ n 8 with r)          k_gcr_axpy   m ldq sizeof(QT) + n 24
//   k_gcr_flush  m ldz 8 + n 16 + knew n 8                      k_gcr_update n (8 + 8 + 16 + 16 + sizeof(QT))
#include "fsi_kernels.hpp"

namespace fsi {

namespace {

__device__ inline double wave_sum_d(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// sum over a 256-thread block; valid in thread 0
__device__ inline double block_sum256(double v) {
  __shared__ double sh[4];
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// NT: the columns of a large store are read once per launch and never again before the whole store has gone by - non-temporal
// loads keep them from displacing w (re-read by every column group) and r in L2 / Infinity Cache (measured on the FP32 store
// of the bench: dots 974 -> 904 us, update 924 -> 840 us per launch).  Not for the FP64 window: its few columns are read by
// the product kernel and again by the update kernel right behind it, out of the Infinity Cache (213 -> 247 us with NT).
template <bool NT>
__device__ inline void load4(const float* p, double& a, double& b, double& c, double& d) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(p)) : *reinterpret_cast<const f4*>(p);
  a = (double)v.x; b = (double)v.y; c = (double)v.z; d = (double)v.w;
}
template <bool NT>
__device__ inline void load4(const double* p, double& a, double& b, double& c, double& d) {
  if (NT) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* q = reinterpret_cast<const d2*>(p);
    const d2 u = __builtin_nontemporal_load(q), v = __builtin_nontemporal_load(q + 1);
    a = u.x; b = u.y; c = v.x; d = v.y;
  } else {
    const double4 v = *reinterpret_cast<const double4*>(p);
    a = v.x; b = v.y; c = v.z; d = v.w;
  }
}

// part[k * gridDim.x + bx] = partial of Q_k . w (k < m);  k = m: w . w;  k = m + 1: w . r (0 when r == nullptr).
// blockIdx.y selects NC directions (or, in the last row, the two vector products): one read of w feeds NC products, so
// the FP32 form takes eight columns per block (w is FP64: with four, re-reading it would add half of Q's bytes again).
template <class QT, int NC, bool NT>
__global__ __launch_bounds__(256) void k_gcr_dots(const QT* __restrict__ Q, int64_t ldq, int64_t n, int m,
                                                  const double* __restrict__ w, const double* __restrict__ r,
                                                  double* __restrict__ part) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int ngroups = (m + NC - 1) / NC;
  if ((int)blockIdx.y == ngroups) {
    double sww = 0.0, swr = 0.0;
    for (int64_t i = t0; i < n; i += stride) {
      const double wv = w[i];
      sww += wv * wv;
      if (r) swr += wv * r[i];
    }
    sww = block_sum256(sww);
    swr = block_sum256(swr);
    if (threadIdx.x == 0) {
      part[(int64_t)m * gridDim.x + blockIdx.x] = sww;
      part[(int64_t)(m + 1) * gridDim.x + blockIdx.x] = swr;
    }
    return;
  }
  const int k0 = NC * blockIdx.y;
  const QT* q[NC];
  double s[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { q[c] = Q + (int64_t)(k0 + c < m ? k0 + c : k0) * ldq; s[c] = 0.0; }
  for (int64_t t = t0; t < n4; t += stride) {
    const int64_t i = t << 2;
    const double2 wa = *reinterpret_cast<const double2*>(w + i);
    const double2 wb = *reinterpret_cast<const double2*>(w + i + 2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double a, b, cc, d;
      load4<NT>(q[c] + i, a, b, cc, d);
      s[c] += (a * wa.x + b * wa.y) + (cc * wb.x + d * wb.y);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {       // the last n mod 4 entries
    const int64_t i = (n4 << 2) + threadIdx.x;
    const double wv = w[i];
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] += (double)q[c][i] * wv;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double v = block_sum256(s[c]);
    if (threadIdx.x == 0 && k0 + c < m) part[(int64_t)(k0 + c) * gridDim.x + blockIdx.x] = v;
  }
}

// block b sums part[b * np .. b * np + np)
__global__ __launch_bounds__(256) void k_gcr_sum(const double* __restrict__ part, int np, double* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += part[(int64_t)blockIdx.x * np + i];
  s = block_sum256(s);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// w -= sum_k h[k] Q_k;  part[bx] = partial |w'|^2,  part[gridDim.x + bx] = partial w'.r
template <class QT, bool NT>
__global__ __launch_bounds__(256) void k_gcr_axpy(const QT* __restrict__ Q, int64_t ldq, int64_t n, int m,
                                                  const double* __restrict__ h, double* __restrict__ w,
                                                  const double* __restrict__ r, double* __restrict__ part) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double sww = 0.0, swr = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n4; t += stride) {
    const int64_t i = t << 2;
    const QT* q = Q + i;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int k = 0;
    for (; k + 4 <= m; k += 4) {            // four independent 16-byte (FP32) / 32-byte (FP64) loads in flight
      double x0, x1, x2, x3, y0, y1, y2, y3, z0, z1, z2, z3, u0, u1, u2, u3;
      load4<NT>(q + (int64_t)k * ldq, x0, x1, x2, x3);
      load4<NT>(q + (int64_t)(k + 1) * ldq, y0, y1, y2, y3);
      load4<NT>(q + (int64_t)(k + 2) * ldq, z0, z1, z2, z3);
      load4<NT>(q + (int64_t)(k + 3) * ldq, u0, u1, u2, u3);
      const double h0 = h[k], h1 = h[k + 1], h2 = h[k + 2], h3 = h[k + 3];
      a0 += h0 * x0 + h2 * z0; a1 += h0 * x1 + h2 * z1; a2 += h0 * x2 + h2 * z2; a3 += h0 * x3 + h2 * z3;
      b0 += h1 * y0 + h3 * u0; b1 += h1 * y1 + h3 * u1; b2 += h1 * y2 + h3 * u2; b3 += h1 * y3 + h3 * u3;
    }
    for (; k < m; ++k) {
      double x0, x1, x2, x3;
      load4<NT>(q + (int64_t)k * ldq, x0, x1, x2, x3);
      const double h0 = h[k];
      a0 += h0 * x0; a1 += h0 * x1; a2 += h0 * x2; a3 += h0 * x3;
    }
    double2 wa = *reinterpret_cast<double2*>(w + i), wb = *reinterpret_cast<double2*>(w + i + 2);
    wa.x -= a0 + b0; wa.y -= a1 + b1; wb.x -= a2 + b2; wb.y -= a3 + b3;
    *reinterpret_cast<double2*>(w + i) = wa;
    *reinterpret_cast<double2*>(w + i + 2) = wb;
    sww += (wa.x * wa.x + wa.y * wa.y) + (wb.x * wb.x + wb.y * wb.y);
    if (r) {
      const double2 ra = *reinterpret_cast<const double2*>(r + i), rb = *reinterpret_cast<const double2*>(r + i + 2);
      swr += (wa.x * ra.x + wa.y * ra.y) + (wb.x * rb.x + wb.y * rb.y);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    double a = 0.0;
    for (int k = 0; k < m; ++k) a += h[k] * (double)Q[(int64_t)k * ldq + i];
    const double wv = w[i] - a;
    w[i] = wv;
    sww += wv * wv;
    if (r) swr += wv * r[i];
  }
  sww = block_sum256(sww);
  swr = block_sum256(swr);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = sww;
    part[gridDim.x + blockIdx.x] = swr;
  }
}

// q = w * inv_wn  ->  Q_slot (rounded to QT) and qd (FP64: the vector the next direction is made from), raw z -> Z_slot,
// r -= alpha q;  part[bx] = partial |r'|^2
template <class QT>
__global__ __launch_bounds__(256) void k_gcr_update(QT* __restrict__ qcol, double* __restrict__ zcol, int64_t n,
                                                    const double* __restrict__ w, const double* __restrict__ z,
                                                    double inv_wn, double alpha, double* __restrict__ r,
                                                    double* __restrict__ qd, double* __restrict__ part) {
  double srr = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const double q = w[i] * inv_wn;
    qcol[i] = (QT)q;
    qd[i] = q;
    zcol[i] = z[i];
    const double rv = r[i] - alpha * q;
    r[i] = rv;
    srr += rv * rv;
  }
  srr = block_sum256(srr);
  if (threadIdx.x == 0) part[blockIdx.x] = srr;
}

// One pass over the direction store: x += sum_j y[j] Z_j, and for the KN directions made since the last pass
// p_k = sum_j cn[k * m + j] Z_j written over their raw vectors (row i of the result only needs row i of Z, so the
// overwrite is in place).  Two rows per thread (double2 loads).
template <int KN>
__global__ __launch_bounds__(256) void k_gcr_flush(double* __restrict__ Z, int64_t ldz, int64_t n, int m,
                                                   const double* __restrict__ y, const double* __restrict__ cn,
                                                   const int32_t* __restrict__ slots, int knew, double* __restrict__ x) {
  const int64_t n2 = (n + 1) >> 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n2; t += stride) {
    const int64_t i = t << 1;
    double ax0 = 0.0, ax1 = 0.0;
    double p0[KN > 0 ? KN : 1], p1[KN > 0 ? KN : 1];
#pragma unroll
    for (int k = 0; k < KN; ++k) { p0[k] = 0.0; p1[k] = 0.0; }
    const double* zc = Z + i;
#pragma unroll 4      // (8: 3.02 -> 3.11 ms per launch)
    for (int j = 0; j < m; ++j) {
      const double2 v = *reinterpret_cast<const double2*>(zc + (int64_t)j * ldz);      // ldz is even, columns padded
      const double yj = y[j];
      ax0 += yj * v.x; ax1 += yj * v.y;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const double c = cn[(int64_t)k * m + j];
        p0[k] += c * v.x; p1[k] += c * v.y;
      }
    }
    const bool two = i + 1 < n;
    x[i] += ax0;
    if (two) x[i + 1] += ax1;
#pragma unroll
    for (int k = 0; k < KN; ++k)
      if (k < knew) {
        double* zo = Z + (int64_t)slots[k] * ldz + i;
        zo[0] = p0[k];
        if (two) zo[1] = p1[k];
      }
  }
}

// columns that do not fit the Infinity Cache (256 MiB) beside w and r are gone before the next pass comes by: stream them
#ifndef FSI_GCR_STREAM_MIB
#define FSI_GCR_STREAM_MIB 512.0
#endif
template <class QT>
inline bool stream_once(int64_t ldq, int m) {      // (FP32 columns only: an FP64 store came out 8 % slower with them, 1.83 -> 1.98 s of 20 bench steps)
  return sizeof(QT) == 4 && (double)ldq * (double)m * sizeof(QT) > FSI_GCR_STREAM_MIB * 1048576.0;
}
template <class QT>
void dots_t(hipStream_t st, const void* Q, int64_t ldq, int64_t n, int m, const double* w, const double* r,
            double* scratch, double* out) {
  // row parts: 128 at a full store (3 000 workgroups at m = 180; 64: 4 % slower, 256: same) and more when there are few columns - the
  // FP64 window's 1 - 8 columns were 256 workgroups on 256 CUs (212 us for 0.4 GB) - so that the grid has ~2 000 workgroups either way
  const int ngroups0 = (m + 7) / 8 + 1;
  int npmax = std::min(1024, std::max(128, (2048 + ngroups0 - 1) / ngroups0));
  int np = (int)((n + 4095) / 4096);      // at least 4 096 rows per part
  if (np < 1) np = 1;
  if (np > npmax) np = npmax;
  // eight columns per read of w, FP32 and FP64 columns alike: w is re-read for 1/8 of Q's bytes (FP64 with four: 1 930 -> 1 898 ms per
  // 20 bench steps)
  const int ngroups = (m + 7) / 8;
  if (stream_once<QT>(ldq, m))
    hipLaunchKernelGGL((k_gcr_dots<QT, 8, true>), dim3(np, ngroups + 1), dim3(256), 0, st, static_cast<const QT*>(Q), ldq, n, m, w, r, scratch);
  else
    hipLaunchKernelGGL((k_gcr_dots<QT, 8, false>), dim3(np, ngroups + 1), dim3(256), 0, st, static_cast<const QT*>(Q), ldq, n, m, w, r, scratch);
  hipLaunchKernelGGL(k_gcr_sum, dim3(m + 2), dim3(256), 0, st, scratch, np, out);
}
template <class QT>
void axpy_t(hipStream_t st, const void* Q, int64_t ldq, int64_t n, int m, const double* h, double* w, const double* r,
            double* scratch, double* out2) {
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;               // 1024 ... 8192 measured: no difference
  if (stream_once<QT>(ldq, m))
    hipLaunchKernelGGL((k_gcr_axpy<QT, true>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const QT*>(Q), ldq, n, m, h, w, r, scratch);
  else
    hipLaunchKernelGGL((k_gcr_axpy<QT, false>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const QT*>(Q), ldq, n, m, h, w, r, scratch);
  hipLaunchKernelGGL(k_gcr_sum, dim3(2), dim3(256), 0, st, scratch, (int)blocks, out2);
}
template <class QT>
void update_t(hipStream_t st, void* Q, int64_t ldq, double* Z, int64_t ldz, int slot, int64_t n, const double* w,
              const double* z, double inv_wn, double alpha, double* r, double* qd, double* scratch, double* out1) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_gcr_update<QT>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<QT*>(Q) + (int64_t)slot * ldq,
                     Z + (int64_t)slot * ldz, n, w, z, inv_wn, alpha, r, qd, scratch);
  hipLaunchKernelGGL(k_gcr_sum, dim3(1), dim3(256), 0, st, scratch, (int)blocks, out1);
}

}  // namespace

void launch_gcr_dots(hipStream_t st, bool fp32, const void* Q, int64_t ldq, int64_t n, int m, const double* w,
                     const double* r, double* scratch, double* out) {
  if (fp32) dots_t<float>(st, Q, ldq, n, m, w, r, scratch, out);
  else dots_t<double>(st, Q, ldq, n, m, w, r, scratch, out);
}
void launch_gcr_axpy(hipStream_t st, bool fp32, const void* Q, int64_t ldq, int64_t n, int m, const double* h, double* w,
                     const double* r, double* scratch, double* out2) {
  if (fp32) axpy_t<float>(st, Q, ldq, n, m, h, w, r, scratch, out2);
  else axpy_t<double>(st, Q, ldq, n, m, h, w, r, scratch, out2);
}
void launch_gcr_update(hipStream_t st, bool fp32, void* Q, int64_t ldq, double* Z, int64_t ldz, int slot, int64_t n,
                       const double* w, const double* z, double inv_wn, double alpha, double* r, double* qd,
                       double* scratch, double* out1) {
  if (fp32) update_t<float>(st, Q, ldq, Z, ldz, slot, n, w, z, inv_wn, alpha, r, qd, scratch, out1);
  else update_t<double>(st, Q, ldq, Z, ldz, slot, n, w, z, inv_wn, alpha, r, qd, scratch, out1);
}
int gcr_flush_width(int knew) { return knew <= 0 ? 0 : knew <= 4 ? 4 : knew <= 8 ? 8 : knew <= 16 ? 16 : 32; }
void launch_gcr_flush(hipStream_t st, double* Z, int64_t ldz, int64_t n, int m, const double* y, const double* cn,
                      const int32_t* slots, int knew, double* x) {
  if (m <= 0) return;
  int64_t blocks = (((n + 1) >> 1) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 g((unsigned)blocks), b(256);
  switch (gcr_flush_width(knew)) {
    case 0: hipLaunchKernelGGL(k_gcr_flush<0>, g, b, 0, st, Z, ldz, n, m, y, cn, slots, knew, x); break;
    case 4: hipLaunchKernelGGL(k_gcr_flush<4>, g, b, 0, st, Z, ldz, n, m, y, cn, slots, knew, x); break;
    case 8: hipLaunchKernelGGL(k_gcr_flush<8>, g, b, 0, st, Z, ldz, n, m, y, cn, slots, knew, x); break;
    case 16: hipLaunchKernelGGL(k_gcr_flush<16>, g, b, 0, st, Z, ldz, n, m, y, cn, slots, knew, x); break;
    default: hipLaunchKernelGGL(k_gcr_flush<32>, g, b, 0, st, Z, ldz, n, m, y, cn, slots, knew, x); break;
  }
}

}  // namespace fsi
