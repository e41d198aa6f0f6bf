// Third level of the solid velocity block's multilevel cycle: smoothed aggregation with a DENSE coarsest operator.
//
// Where it sits: the solid block of the velocity predictor is solved by a P2 -> P1 two-level cycle
// (fsi_block.hip, k_sbmg_*).  Its P1 level - 3x3 blocks on the solid vertices, a thin-walled, nearly incompressible shell
// - was "solved" by 200 block-Jacobi Chebyshev sweeps: 240 of the ~340 launches of a preconditioner application together
// with the displacement block's coarse level, 11 % of the GPU time, and still far from a solve (an
// energy-norm error of 0.6 on a smooth load: the condition number of that shell is ~1e6, not the 4 000 the interval assumes).
// This file replaces those sweeps by a few two-grid cycles on the P1 level whose coarse space is
//
//   aggregates of ~48 solid vertices (recursive coordinate bisection: patches of the wall), six rigid-body modes each,
//   tentative prolongator smoothed by `deg` damped block-Jacobi steps (smoothed aggregation), blocks below 1e-3 dropped;
//
// ~450 aggregates x 6 = 2 700 unknowns at 1.12 M tets, few enough for an EXPLICIT inverse: the coarse solve is one dense
// matrix-vector product (36 MB in FP32, on-die) instead of a chain of dependent sparse sweeps.
//
// The prolongator is built ONCE, on the host, from the first Jacobian of the context (the wall's tangent stiffness moves by
// |grad d| ~ 1e-3 over a run, so the smoothed basis stays good) and then frozen; every Jacobian refresh recomputes only the
// Galerkin operator A3 = P^T A2 P on the device (two atomic passes through host-made index lists, FP64 accumulation) and its
// inverse (blocked Gauss-Jordan in FP64, 32-wide panels, three launches per panel).  A frozen P keeps A3 the exact Galerkin
// operator of the current matrix, so the cycle stays a convergent symmetric correction whatever P is.
//
// Per application (launches): restrict r3 = P^T r2 (1), x3 = A3^-1 r3 (1), prolongate into the sweep's direction buffer (1).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <vector>

#include "fsi_kernels.hpp"

namespace fsi {

#define HIPCHK(call)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
      return FSI_ERR_DEVICE;                                                                       \
    }                                                                                              \
  } while (0)

namespace {

constexpr int GJ = 32;      // panel width of the blocked Gauss-Jordan inversion

template <class T>
int up(FsiCtx* ctx, DevBuf<T>& buf, const std::vector<T>& h) {
  HIPCHK(buf.alloc(h.size()));
  if (!h.empty()) HIPCHK(hipMemcpy(buf.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return FSI_OK;
}

// ---- Galerkin product, pass 1: T = A2 P.  One thread per (block entry e = (i, j) of A2, block b of P's row j): the 3x6
// product goes to its slot of row i of T (slots made on the host: the union of the aggregates of i's neighbours).
__global__ __launch_bounds__(256) void k_l3_ap(int64_t ntrip, const int32_t* __restrict__ trip_e, const int32_t* __restrict__ trip_b,
                                               const int32_t* __restrict__ trip_t, const float* __restrict__ avals,
                                               const float* __restrict__ pval, float* __restrict__ T) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < ntrip; k += (int64_t)gridDim.x * blockDim.x) {
    const float* a = avals + 9 * (int64_t)trip_e[k];
    const float* p = pval + 18 * (int64_t)trip_b[k];
    float* t = T + 18 * (int64_t)trip_t[k];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int m = 0; m < 6; ++m) unsafeAtomicAdd(&t[6 * c + m], a[3 * c] * p[m] + a[3 * c + 1] * p[6 + m] + a[3 * c + 2] * p[12 + m]);
  }
}
// pass 2: A3 += P_i^T T_i.  One thread per (block b of P's row i, slot s of T's row i): a 6x6 block of the dense matrix.
__global__ __launch_bounds__(256) void k_l3_ptap(int64_t npair, const int32_t* __restrict__ pair_b, const int32_t* __restrict__ pair_t,
                                                 const int32_t* __restrict__ pcol, const int32_t* __restrict__ tcol,
                                                 const float* __restrict__ pval, const float* __restrict__ T, int64_t ld,
                                                 double* __restrict__ A3) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < npair; k += (int64_t)gridDim.x * blockDim.x) {
    const int32_t b = pair_b[k], s = pair_t[k];
    const float* p = pval + 18 * (int64_t)b;
    const float* t = T + 18 * (int64_t)s;
    double* out = A3 + (6 * (int64_t)pcol[b]) * ld + 6 * (int64_t)tcol[s];
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int q = 0; q < 6; ++q)
        unsafeAtomicAdd(&out[m * ld + q], (double)p[m] * t[q] + (double)p[6 + m] * t[6 + q] + (double)p[12 + m] * t[12 + q]);
  }
}
__global__ void k_l3_diag(int64_t n, int64_t ld, const double* __restrict__ A, double* __restrict__ d) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = A[i * ld + i];
}
// rows / columns of dead unknowns (no free vertex carries the mode; padding) become identity rows, and the matrix is made
// exactly symmetric (the two atomic passes add in different orders above and below the diagonal)
__global__ void k_l3_mask(int64_t n, const uint8_t* __restrict__ dead, double* __restrict__ A) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n * n; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = k / n, j = k - i * n;
    if (j < i) continue;
    double v = 0.5 * (A[i * n + j] + A[j * n + i]);
    if (dead[i] || dead[j]) v = (i == j) ? 1.0 : 0.0;
    A[i * n + j] = v;
    A[j * n + i] = v;
  }
}

// ---- blocked Gauss-Jordan inversion in place (SPD matrix, no pivoting), panel kb --------------------------------------------
// step 1: inverse of the 32x32 pivot block (one workgroup, unblocked Gauss-Jordan in LDS)
__global__ __launch_bounds__(1024) void k_gj_pivot(int64_t n, int kb, const double* __restrict__ A, double* __restrict__ dinv) {
  __shared__ double D[GJ][GJ + 1];
  const int i = threadIdx.y, j = threadIdx.x;
  D[i][j] = A[((int64_t)kb * GJ + i) * n + (int64_t)kb * GJ + j];
  __syncthreads();
  for (int k = 0; k < GJ; ++k) {
    const double p = D[k][k], rk = D[k][j], f = D[i][k];
    __syncthreads();
    double v;
    if (i == k) v = (j == k) ? 1.0 / p : rk / p;
    else v = (j == k) ? -f / p : D[i][j] - f * rk / p;
    D[i][j] = v;
    __syncthreads();
  }
  dinv[i * GJ + j] = D[i][j];
}
// step 2: row panel R = Dinv A[kb, :] (32 x n) and a copy of the column panel C = A[:, kb] (n x 32); tile t of both
__global__ __launch_bounds__(1024) void k_gj_panels(int64_t n, int kb, const double* __restrict__ A, const double* __restrict__ dinv,
                                                    double* __restrict__ R, double* __restrict__ Cp) {
  __shared__ double Dv[GJ][GJ + 1], At[GJ][GJ + 1];
  const int i = threadIdx.y, j = threadIdx.x;
  const int64_t t = blockIdx.x;
  Dv[i][j] = dinv[i * GJ + j];
  At[i][j] = A[((int64_t)kb * GJ + i) * n + t * GJ + j];
  Cp[(t * GJ + i) * GJ + j] = A[(t * GJ + i) * n + (int64_t)kb * GJ + j];
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < GJ; ++k) s += Dv[i][k] * At[k][j];
  R[(int64_t)i * n + t * GJ + j] = s;
}
// step 3: A[ib, jb] -= C[ib] R[:, jb] off the panel; the pivot row becomes R, the pivot column -C Dinv, the pivot block Dinv
__global__ __launch_bounds__(256) void k_gj_update(int64_t n, int kb, double* __restrict__ A, const double* __restrict__ dinv,
                                                   const double* __restrict__ R, const double* __restrict__ Cp) {
  __shared__ double Cs[GJ][GJ + 1], Rs[GJ][GJ + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows of 32 threads; each thread four rows of the tile
  const int64_t ib = blockIdx.y, jb = blockIdx.x;
  const bool prow = ib == kb, pcol = jb == kb;
  if (prow) {                                                   // A[kb, jb] = R (or Dinv on the pivot block)
    for (int r = ty; r < GJ; r += 8)
      A[((int64_t)kb * GJ + r) * n + jb * GJ + tx] = pcol ? dinv[r * GJ + tx] : R[(int64_t)r * n + jb * GJ + tx];
    return;
  }
  for (int r = ty; r < GJ; r += 8) {
    Cs[r][tx] = Cp[(ib * GJ + r) * GJ + tx];
    Rs[r][tx] = pcol ? dinv[r * GJ + tx] : R[(int64_t)r * n + jb * GJ + tx];
  }
  __syncthreads();
  for (int r = ty; r < GJ; r += 8) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < GJ; ++k) s += Cs[r][k] * Rs[k][tx];
    double* a = &A[(ib * GJ + r) * n + jb * GJ + tx];
    *a = pcol ? -s : *a - s;
  }
}
__global__ void k_l3_to_f32(int64_t n, const double* __restrict__ a, float* __restrict__ b) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = (float)a[i];
}

// ---- per application -----------------------------------------------------------------------------------------------------
// r3[6 I + m] = sum over the blocks (v, B) of aggregate I's column of P:  B[:, m] . r2[v]   (one wave per aggregate)
__global__ __launch_bounds__(64) void k_l3_restrict(const int64_t* __restrict__ tptr, const int32_t* __restrict__ tvert,
                                                    const int32_t* __restrict__ tblk, const float* __restrict__ pval,
                                                    const float* __restrict__ r4, float* __restrict__ r3) {
  const int64_t I = blockIdx.x;
  float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t k = tptr[I] + threadIdx.x; k < tptr[I + 1]; k += 64) {
    const float4 rv = reinterpret_cast<const float4*>(r4)[tvert[k]];
    const float* p = pval + 18 * (int64_t)tblk[k];
#pragma unroll
    for (int m = 0; m < 6; ++m) s[m] += p[m] * rv.x + p[6 + m] * rv.y + p[12 + m] * rv.z;
  }
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    float v = group_sum<16>(s[m]);
    v = __shfl(v, 0, 64) + __shfl(v, 16, 64) + __shfl(v, 32, 64) + __shfl(v, 48, 64);
    if (threadIdx.x == 0) r3[6 * I + m] = v;
  }
}
// x3 = Ainv r3: one wave per row, float4 loads (n is a multiple of 32)
__global__ __launch_bounds__(256) void k_l3_gemv(int64_t n, const float* __restrict__ Ainv, const float* __restrict__ r3,
                                                 float* __restrict__ x3) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  const float4* a = reinterpret_cast<const float4*>(Ainv + row * n);
  const float4* r = reinterpret_cast<const float4*>(r3);
  float s = 0.f;
  for (int64_t k = lane; k < (n >> 2); k += 64) {
    const float4 av = a[k], rv = r[k];
    s += (av.x * rv.x + av.y * rv.y) + (av.z * rv.z + av.w * rv.w);
  }
  s = group_sum<16>(s);
  s = __shfl(s, 0, 64) + __shfl(s, 16, 64) + __shfl(s, 32, 64) + __shfl(s, 48, 64);
  if (lane == 0) x3[row] = s;
}
// e2[v] = sum over the blocks (I, B) of P's row v:  B x3[6 I ..]   (float4 per vertex, the sweep's direction layout)
__global__ __launch_bounds__(256) void k_l3_prolong(int64_t nc, const int64_t* __restrict__ pptr, const int32_t* __restrict__ pcol,
                                                    const float* __restrict__ pval, const float* __restrict__ x3,
                                                    float* __restrict__ e4) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nc; v += (int64_t)gridDim.x * blockDim.x) {
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
    for (int64_t b = pptr[v]; b < pptr[v + 1]; ++b) {
      const float* p = pval + 18 * b;
      const float* x = x3 + 6 * (int64_t)pcol[b];
#pragma unroll
      for (int m = 0; m < 6; ++m) { e0 += p[m] * x[m]; e1 += p[6 + m] * x[m]; e2 += p[12 + m] * x[m]; }
    }
    reinterpret_cast<float4*>(e4)[v] = make_float4(e0, e1, e2, 0.f);
  }
}

unsigned grid_for(int64_t n, int cap = 16384) {
  int64_t b = (n + 255) / 256;
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(b, cap));
}

using Blk = std::array<double, 18>;           // 3 x 6, row-major
using Row = std::map<int32_t, Blk>;           // aggregate -> block

void rcb(const std::vector<double>& xyz, std::vector<int32_t>& idx, int64_t lo, int64_t hi, int maxsize, int32_t& next,
         std::vector<int32_t>& agg) {
  if (hi - lo <= maxsize) {
    for (int64_t k = lo; k < hi; ++k) agg[idx[k]] = next;
    next += 1;
    return;
  }
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int64_t k = lo; k < hi; ++k)
    for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], xyz[3 * (size_t)idx[k] + c]); mx[c] = std::max(mx[c], xyz[3 * (size_t)idx[k] + c]); }
  int ax = 0;
  for (int c = 1; c < 3; ++c) if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;
  const int64_t mid = lo + (hi - lo) / 2;
  std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int32_t a, int32_t b) {
    const double xa = xyz[3 * (size_t)a + ax], xb = xyz[3 * (size_t)b + ax];
    return xa < xb || (xa == xb && a < b);
  });
  rcb(xyz, idx, lo, mid, maxsize, next, agg);
  rcb(xyz, idx, mid, hi, maxsize, next, agg);
}

}  // namespace

// One-off host construction of the frozen prolongator from the P1-level matrix that is on the device right now.
int l3_build(FsiCtx* ctx) {
  L3Level& L = ctx->l3;
  const int64_t nc = ctx->sbmg_nc, nblk = ctx->sbmg_nblk;
  if (const char* e = getenv("FSI_L3_AGG")) L.aggsize = std::max(4, atoi(e));
  if (const char* e = getenv("FSI_L3_DEG")) L.deg = std::max(0, atoi(e));
  if (const char* e = getenv("FSI_L3_PRE")) L.pre = std::max(0, atoi(e));
  if (const char* e = getenv("FSI_L3_POST")) L.post = std::max(1, atoi(e));
  if (const char* e = getenv("FSI_L3_CYCLES")) L.cycles = std::max(1, atoi(e));
  if (const char* e = getenv("FSI_L3_ALPHA")) L.alpha = atof(e);
  if (const char* e = getenv("FSI_L3_TRUE_LMAX")) L.true_lmax = atoi(e);
  L.built = true;                       // one attempt per context, whatever comes of it
  if (nc < 4 * L.aggsize || ctx->h_sc_xyz.size() != 3 * (size_t)nc) return FSI_OK;      // too small to be worth a level
  std::vector<float> cvals(9 * (size_t)nblk), binv(12 * (size_t)nc);
  std::vector<uint8_t> cflag(nc);
  HIPCHK(hipMemcpy(cvals.data(), ctx->sbmg_cvals.p, cvals.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(binv.data(), ctx->sbmg_cbinv12.p, binv.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cflag.data(), ctx->sbmg_cflag.p, nc, hipMemcpyDeviceToHost));
  const std::vector<int64_t>& ptr = ctx->h_sc_ptr;
  const std::vector<int32_t>& col = ctx->h_sc_col;
  const std::vector<double>& xyz = ctx->h_sc_xyz;
  // aggregates: recursive coordinate bisection of the solid vertices (patches of the wall)
  std::vector<int32_t> idx(nc), agg(nc, -1);
  std::iota(idx.begin(), idx.end(), 0);
  int32_t nagg = 0;
  rcb(xyz, idx, 0, nc, L.aggsize, nagg, agg);
  // tentative prolongator: per vertex [I | -[x - c]x / R], zero rows for identity (Dirichlet / ghost) vertices
  std::vector<double> cen(3 * (size_t)nagg, 0.0), rad(nagg, 0.0);
  std::vector<int32_t> cnt(nagg, 0);
  for (int64_t v = 0; v < nc; ++v) { cnt[agg[v]] += 1; for (int c = 0; c < 3; ++c) cen[3 * (size_t)agg[v] + c] += xyz[3 * v + c]; }
  for (int32_t I = 0; I < nagg; ++I) for (int c = 0; c < 3; ++c) cen[3 * (size_t)I + c] /= std::max(1, cnt[I]);
  for (int64_t v = 0; v < nc; ++v) { double s = 0; for (int c = 0; c < 3; ++c) { const double d = xyz[3 * v + c] - cen[3 * (size_t)agg[v] + c]; s += d * d; } rad[agg[v]] += s; }
  for (int32_t I = 0; I < nagg; ++I) rad[I] = std::sqrt(rad[I] / std::max(1, cnt[I])) + 1e-300;
  // an aggregate whose free vertices are (nearly) collinear, or fewer than three, cannot carry three rotations: their
  // columns would be linearly dependent and the coarse matrix singular.  R = sum (|d|^2 I - d d^T) over the free vertices
  // is the rotation block of the tentative Gram matrix; its smallest eigenvalue decides.
  std::vector<uint8_t> norot(nagg, 0);
  {
    std::vector<std::array<double, 9>> R(nagg, std::array<double, 9>{});
    std::vector<int32_t> nfree(nagg, 0);
    for (int64_t v = 0; v < nc; ++v) {
      if (cflag[v]) continue;
      const int32_t I = agg[v];
      double d[3];
      for (int c = 0; c < 3; ++c) d[c] = (xyz[3 * v + c] - cen[3 * (size_t)I + c]) / rad[I];
      const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) R[I][3 * a + b] += (a == b ? d2 : 0.0) - d[a] * d[b];
      nfree[I] += 1;
    }
    for (int32_t I = 0; I < nagg; ++I) {
      // smallest eigenvalue of the symmetric 3x3 by a few cyclic Jacobi rotations
      double a[3][3];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a[r][c] = R[I][3 * r + c];
      for (int sweep = 0; sweep < 8; ++sweep)
        for (int p_ = 0; p_ < 2; ++p_)
          for (int q = p_ + 1; q < 3; ++q) {
            if (std::fabs(a[p_][q]) < 1e-300) continue;
            const double th = 0.5 * std::atan2(2.0 * a[p_][q], a[q][q] - a[p_][p_]), cs = std::cos(th), sn = std::sin(th);
            for (int k = 0; k < 3; ++k) { const double x = a[k][p_], y = a[k][q]; a[k][p_] = cs * x - sn * y; a[k][q] = sn * x + cs * y; }
            for (int k = 0; k < 3; ++k) { const double x = a[p_][k], y = a[q][k]; a[p_][k] = cs * x - sn * y; a[q][k] = sn * x + cs * y; }
          }
      const double emin = std::min(a[0][0], std::min(a[1][1], a[2][2])), tr = a[0][0] + a[1][1] + a[2][2];
      norot[I] = nfree[I] < 3 || !(emin > 1e-3 * tr);
    }
  }
  std::vector<Row> P(nc);
  for (int64_t v = 0; v < nc; ++v) {
    if (cflag[v]) continue;
    const int32_t I = agg[v];
    const double dx = (xyz[3 * v] - cen[3 * (size_t)I]) / rad[I], dy = (xyz[3 * v + 1] - cen[3 * (size_t)I + 1]) / rad[I],
                 dz = (xyz[3 * v + 2] - cen[3 * (size_t)I + 2]) / rad[I];
    Blk b{};
    b[0] = 1; b[7] = 1; b[14] = 1;                     // translations
    if (!norot[I]) {
      b[4] = dz; b[5] = -dy;                           // row x: rotations about (x, y, z) -> (0, z, -y)
      b[6 + 3] = -dz; b[6 + 5] = dx;                   // row y: (-z, 0, x)
      b[12 + 3] = dy; b[12 + 4] = -dx;                 // row z: (y, -x, 0)
    }
    P[v][I] = b;
  }
  // largest eigenvalue of B^-1 A on this level by power iteration (the level's sweeps use a Gershgorin bound, several
  // times larger; a damping 4 / (3 bound) would leave the prolongator almost unsmoothed)
  double lmax = ctx->sbmg_clmax;
  {
    std::vector<double> x(3 * (size_t)nc), y(3 * (size_t)nc);
    for (int64_t i = 0; i < 3 * nc; ++i) x[i] = cflag[i / 3] ? 0.0 : ((i * 2654435761u) % 1000) / 500.0 - 1.0;
    double lam = 0.0;
    for (int it = 0; it < 30; ++it) {
      double nx = 0.0, ny = 0.0;
      for (int64_t i = 0; i < nc; ++i) {
        double t[3] = {0, 0, 0};
        if (!cflag[i])
          for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e) {
            const float* a = &cvals[9 * (size_t)e];
            const double* xj = &x[3 * (size_t)col[e]];
            for (int c = 0; c < 3; ++c) t[c] += a[3 * c] * xj[0] + a[3 * c + 1] * xj[1] + a[3 * c + 2] * xj[2];
          }
        const float* bi = &binv[12 * (size_t)i];
        for (int c = 0; c < 3; ++c) y[3 * i + c] = cflag[i] ? 0.0 : bi[4 * c] * t[0] + bi[4 * c + 1] * t[1] + bi[4 * c + 2] * t[2];
      }
      for (int64_t i = 0; i < 3 * nc; ++i) { nx += x[i] * x[i]; ny += y[i] * y[i]; }
      if (!(nx > 0.0) || !(ny > 0.0)) break;
      lam = std::sqrt(ny / nx);
      const double s = 1.0 / std::sqrt(ny);
      for (int64_t i = 0; i < 3 * nc; ++i) x[i] = y[i] * s;
    }
    if (lam > 0.0 && std::isfinite(lam)) lmax = std::min(lmax, 1.1 * lam);
  }
  L.lmax = lmax;
  // prolongator smoothing: P <- (I - w B^-1 A) P, w = 4 / (3 lmax), with the block-Jacobi scaling of the level's sweeps
  const double w = (4.0 / 3.0) / std::max(1e-30, lmax);
  for (int s = 0; s < L.deg; ++s) {
    std::vector<Row> Q(nc);
    for (int64_t i = 0; i < nc; ++i) {
      if (cflag[i]) continue;
      Row acc;
      for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e) {
        const int32_t j = col[e];
        if (P[j].empty()) continue;
        const float* a = &cvals[9 * (size_t)e];
        for (const auto& kv : P[j]) {
          Blk& t = acc[kv.first];
          for (int c = 0; c < 3; ++c)
            for (int m = 0; m < 6; ++m) t[6 * c + m] += a[3 * c] * kv.second[m] + a[3 * c + 1] * kv.second[6 + m] + a[3 * c + 2] * kv.second[12 + m];
        }
      }
      Row out = P[i];
      const float* bi = &binv[12 * (size_t)i];
      for (const auto& kv : acc) {
        Blk& o = out[kv.first];
        for (int c = 0; c < 3; ++c)
          for (int m = 0; m < 6; ++m)
            o[6 * c + m] -= w * (bi[4 * c] * kv.second[m] + bi[4 * c + 1] * kv.second[6 + m] + bi[4 * c + 2] * kv.second[12 + m]);
      }
      Q[i] = std::move(out);
    }
    P.swap(Q);
  }
  // drop what is negligible in its row, then the two sparse layouts (by vertex, by aggregate)
  std::vector<int64_t> pptr(nc + 1, 0);
  std::vector<int32_t> pcol;
  std::vector<float> pval;
  for (int64_t v = 0; v < nc; ++v) {
    double big = 0.0;
    for (const auto& kv : P[v]) { double f = 0; for (double x : kv.second) f += x * x; big = std::max(big, f); }
    for (const auto& kv : P[v]) {
      double f = 0;
      for (double x : kv.second) f += x * x;
      if (f < 1e-6 * big || f == 0.0) continue;
      pcol.push_back(kv.first);
      for (double x : kv.second) pval.push_back((float)x);
    }
    pptr[v + 1] = (int64_t)pcol.size();
  }
  const int64_t npb = (int64_t)pcol.size();
  std::vector<int64_t> tptr(nagg + 1, 0);
  for (int32_t I : pcol) tptr[I + 1] += 1;
  for (int32_t I = 0; I < nagg; ++I) tptr[I + 1] += tptr[I];
  std::vector<int32_t> tvert(npb), tblk(npb);
  {
    std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
    for (int64_t v = 0; v < nc; ++v)
      for (int64_t b = pptr[v]; b < pptr[v + 1]; ++b) { const int64_t pos = fill[pcol[b]]++; tvert[pos] = (int32_t)v; tblk[pos] = (int32_t)b; }
  }
  // index lists of the Galerkin product: T = A P has, in row i, one 3x6 slot per aggregate reached through i's neighbours
  std::vector<int64_t> trow(nc + 1, 0);
  std::vector<int32_t> tcol, trip_e, trip_b, trip_t, pair_b, pair_t;
  for (int64_t i = 0; i < nc; ++i) {
    std::map<int32_t, int32_t> slot;
    if (pptr[i + 1] > pptr[i])
      for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e)
        for (int64_t b = pptr[col[e]]; b < pptr[col[e] + 1]; ++b) slot.emplace(pcol[b], 0);
    for (auto& kv : slot) { kv.second = (int32_t)tcol.size(); tcol.push_back(kv.first); }
    trow[i + 1] = (int64_t)tcol.size();
    if (slot.empty()) continue;
    for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e)
      for (int64_t b = pptr[col[e]]; b < pptr[col[e] + 1]; ++b) { trip_e.push_back((int32_t)e); trip_b.push_back((int32_t)b); trip_t.push_back(slot[pcol[b]]); }
    for (int64_t b = pptr[i]; b < pptr[i + 1]; ++b)
      for (int64_t s = trow[i]; s < trow[i + 1]; ++s) { pair_b.push_back((int32_t)b); pair_t.push_back((int32_t)s); }
  }
  if (tcol.size() > (size_t)INT32_MAX / 18 || trip_e.size() > (size_t)INT32_MAX) return FSI_OK;
  L.nagg = nagg;
  L.nd = 6 * (int64_t)nagg;
  L.ndp = (L.nd + GJ - 1) / GJ * GJ;
  L.npb = npb;
  L.ntrip = (int64_t)trip_e.size();
  L.npair = (int64_t)pair_b.size();
  L.nslot = (int64_t)tcol.size();
  if (up(ctx, L.pptr, pptr) || up(ctx, L.pcol, pcol) || up(ctx, L.pval, pval) || up(ctx, L.tptr, tptr) || up(ctx, L.tvert, tvert) ||
      up(ctx, L.tblk, tblk) || up(ctx, L.tcol, tcol) || up(ctx, L.trip_e, trip_e) || up(ctx, L.trip_b, trip_b) ||
      up(ctx, L.trip_t, trip_t) || up(ctx, L.pair_b, pair_b) || up(ctx, L.pair_t, pair_t))
    return FSI_ERR_DEVICE;
  HIPCHK(L.T.alloc(18 * (size_t)L.nslot));
  HIPCHK(L.A3.alloc((size_t)L.ndp * L.ndp));
  HIPCHK(L.Ainv.alloc((size_t)L.ndp * L.ndp));
  HIPCHK(L.panelR.alloc((size_t)GJ * L.ndp));
  HIPCHK(L.panelC.alloc((size_t)GJ * L.ndp));
  HIPCHK(L.dinv.alloc(GJ * GJ));
  HIPCHK(L.diag.alloc(L.ndp));
  HIPCHK(L.dead.alloc(L.ndp));
  HIPCHK(L.r3.alloc(L.ndp));
  HIPCHK(L.x3.alloc(L.ndp));
  L.usable = true;
  if (getenv("FSI_DEBUG"))
    fprintf(stderr, "[fsi] solid level 3: %lld vertices -> %d aggregates (%lld unknowns, padded %lld), P %.1f blocks per vertex, "
            "%lld + %lld Galerkin index triples / pairs; lmax(B^-1 A) %.3f (Gershgorin bound %.3f), %d aggregates without rotations\n",
            (long long)nc, nagg, (long long)L.nd, (long long)L.ndp, (double)npb / std::max<int64_t>(1, nc), (long long)L.ntrip,
            (long long)L.npair, L.lmax, ctx->sbmg_clmax, (int)std::count(norot.begin(), norot.end(), (uint8_t)1));
  return FSI_OK;
}

// Every Jacobian refresh: Galerkin operator of the current P1-level matrix and its inverse.
int l3_refresh(FsiCtx* ctx) {
  L3Level& L = ctx->l3;
  L.ready = false;
  if (!L.usable) return FSI_OK;
  hipStream_t st = ctx->stream;
  const int64_t n = L.ndp;
  HIPCHK(hipMemsetAsync(L.T.p, 0, L.T.n * sizeof(float), st));
  HIPCHK(hipMemsetAsync(L.A3.p, 0, L.A3.n * sizeof(double), st));
  hipLaunchKernelGGL(k_l3_ap, dim3(grid_for(L.ntrip)), dim3(256), 0, st, L.ntrip, L.trip_e.p, L.trip_b.p, L.trip_t.p, ctx->sbmg_cvals.p,
                     L.pval.p, L.T.p);
  hipLaunchKernelGGL(k_l3_ptap, dim3(grid_for(L.npair)), dim3(256), 0, st, L.npair, L.pair_b.p, L.pair_t.p, L.pcol.p, L.tcol.p, L.pval.p,
                     L.T.p, n, L.A3.p);
  hipLaunchKernelGGL(k_l3_diag, dim3(grid_for(n)), dim3(256), 0, st, n, n, L.A3.p, L.diag.p);
  std::vector<double> dg(n);
  HIPCHK(hipMemcpyAsync(dg.data(), L.diag.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  double big = 0.0;
  bool finite = true;
  for (int64_t i = 0; i < L.nd; ++i) { finite = finite && std::isfinite(dg[i]); big = std::max(big, dg[i]); }
  if (!finite || !(big > 0.0)) return FSI_OK;
  std::vector<uint8_t> dead(n, 1);
  int64_t ndead = 0;
  for (int64_t i = 0; i < L.nd; ++i) { dead[i] = !(dg[i] > 1e-10 * big); ndead += dead[i]; }
  HIPCHK(hipMemcpyAsync(L.dead.p, dead.data(), n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_l3_mask, dim3(grid_for(n * n, 65536)), dim3(256), 0, st, n, L.dead.p, L.A3.p);
  const int nb = (int)(n / GJ);
  for (int kb = 0; kb < nb; ++kb) {
    hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(GJ, GJ), 0, st, n, kb, L.A3.p, L.dinv.p);
    hipLaunchKernelGGL(k_gj_panels, dim3(nb), dim3(GJ, GJ), 0, st, n, kb, L.A3.p, L.dinv.p, L.panelR.p, L.panelC.p);
    hipLaunchKernelGGL(k_gj_update, dim3(nb, nb), dim3(256), 0, st, n, kb, L.A3.p, L.dinv.p, L.panelR.p, L.panelC.p);
  }
  hipLaunchKernelGGL(k_l3_to_f32, dim3(grid_for(n * n, 65536)), dim3(256), 0, st, n * n, L.A3.p, L.Ainv.p);
  hipLaunchKernelGGL(k_l3_diag, dim3(grid_for(n)), dim3(256), 0, st, n, n, L.A3.p, L.diag.p);
  HIPCHK(hipMemcpyAsync(dg.data(), L.diag.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  // the inverse of an SPD matrix has a positive diagonal: anything else means the elimination broke down (the cycle then
  // falls back to the plain coarse sweeps for this Jacobian)
  bool ok = true;
  for (int64_t i = 0; i < n; ++i) ok = ok && std::isfinite(dg[i]) && dg[i] > 0.0;
  L.ready = ok;
  if (getenv("FSI_DEBUG"))
    fprintf(stderr, "[fsi] solid level 3 refresh: %lld dead unknowns of %lld, inverse %s\n", (long long)ndead, (long long)L.nd, ok ? "ok" : "BROKE DOWN");
  return FSI_OK;
}

// e2 = P A3^-1 P^T r2 into the direction buffer of the next sweep
void l3_correct(FsiCtx* ctx, const float* r4, float* e4) {
  L3Level& L = ctx->l3;
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_l3_restrict, dim3((unsigned)L.nagg), dim3(64), 0, st, L.tptr.p, L.tvert.p, L.tblk.p, L.pval.p, r4, L.r3.p);
  hipLaunchKernelGGL(k_l3_gemv, dim3((unsigned)((L.ndp * 64 + 255) / 256)), dim3(256), 0, st, L.ndp, L.Ainv.p, L.r3.p, L.x3.p);
  hipLaunchKernelGGL(k_l3_prolong, dim3(grid_for(ctx->sbmg_nc)), dim3(256), 0, st, ctx->sbmg_nc, L.pptr.p, L.pcol.p, L.pval.p, L.x3.p, e4);
}

}  // namespace fsi
