// Exact solve of the solid cycle's coarse level (3x3 blocks on the solid vertices) by BLOCK CYCLIC REDUCTION - round 5.
//
// The level was 90 block-Jacobi / Chebyshev sweeps per preconditioner application: 90 dependent launches of 3 - 6 us on on-die
// data (0.28 ms at 140 k tets, 0.52 ms at 1.12 M: the longest single piece of the application's critical chain at the per-GPU
// size of an 8-rank run), and still far from solving a level whose condition number is above 2e4.
// A vessel wall is a thin tube: in breadth-first order from one end the coarse operator is BLOCK TRIDIAGONAL with blocks of
// one "ring" (a BFS level: 72 vertices = 216 unknowns at 140 k tets, 144 = 432 at 1.12 M) - for any mesh, by construction of
// the levels.  Cyclic reduction eliminates every other block per level, so the solve is 2 log2(K) + 1 launches of batched
// dense matrix-vector products with operators that are PRECOMPUTED at every Jacobian refresh:
//     forward,  level l:  b_j += G_jl b_l + G_jr b_r          (j survives, l / r are its eliminated neighbours; G = -T_j. D^-1)
//     top:                x_t  = D_t^-1 b_t
//     backward, level l:  x_e  = D_e^-1 b_e + H_ea x_a + H_ec x_c   (H = -D_e^-1 T_e.)
// Operators are stored in FP32 (5 K m^2 values: 71 MB at 140 k tets, 0.56 GB at 1.12 M - the solve streams them once),
// vectors and every accumulation are FP64; the set-up (dense inverses by a blocked Gauss-Jordan, products on the FP64 matrix
// pipe: v_mfma_f64_16x16x4_f64) is FP64 throughout.  No pivoting across blocks: the level is the Galerkin operator of an
// elasticity + mass matrix whose Schur complements stay definite; a non-finite or vanishing pivot is reported and the cycle
// falls back to its sweeps for that Jacobian.
//
// Built once (bcr_plan, host: BFS levels, reduction schedule, descriptor tables), refreshed with the Jacobian (bcr_refresh),
// applied inside precondition_block (bcr_solve).  The reference has no counterpart: its linear solver is MUMPS
// [REF src/vasp/simulations/offset_stenosis.py:45].
#include "fsi_host.hpp"

#include <queue>

namespace fsi {

struct BcrSeg { int32_t off, len, src; };                    // input segment of a solve task: src 0 = b, 1 = x
struct BcrTask {                                             // out[rows] (+)= W[rows][ldw] . concat(segments)
  int64_t w;                                                 // offset of W in the FP32 arena
  int32_t rows, ldw, out, nseg;
  BcrSeg seg[3];
};
struct BcrTile { int32_t task, row0; };                      // 16 rows of a task: one workgroup
struct BcrGemm {                                             // C = beta C + alpha (A1 B1 + A2 B2), optional FP32 copy
  int64_t a1, b1, a2, b2, c, o32;
  int32_t M, N, K1, K2, lda1, ldb1, lda2, ldb2, ldc, ld32;
  double alpha, beta;
};
struct BcrGemmTile { int32_t task, ti, tj; };                // 64 x 64 tile of C: one workgroup
struct BcrInv {                                              // in-place inverse of an m x m block (+ FP32 copy)
  int64_t a, o32, cb, rb;                                    // cb [m][32], rb [32][m]: column / row panel scratch of the blocked Gauss-Jordan
  int32_t m, ld, ld32;
};
constexpr int BCR_PANEL = 32;

struct BcrRange { int64_t first = 0, count = 0; };
struct BcrLevelHost {
  BcrRange inv, invupd, gemm1, gemm2, fwd, bwd;            // invupd: tiles of the rank-32 updates A += -Cb Rb of the inverses
  int inv_maxm = 0, fwd_maxld = 0, bwd_maxld = 0;
};

struct BcrData {
  bool planned = false, ready = false;
  int64_t nc = 0, n = 0, K = 0;
  int max_block = 0;
  DevBuf<int32_t> pos;                                       // coarse node -> position in BFS-level order
  DevBuf<int64_t> fill_dst;                                  // per sparse 3x3 block: offset of its (0,0) entry in the FP64 arena
  DevBuf<int32_t> fill_ld;
  int64_t nfill = 0, level0_doubles = 0;
  DevBuf<double> arena64, b, x;
  DevBuf<float> arena32;
  DevBuf<BcrTask> tasks;
  DevBuf<BcrTile> tiles;
  DevBuf<BcrGemm> gemms;
  DevBuf<BcrGemmTile> gtiles;
  DevBuf<BcrInv> invs;
  DevBuf<int32_t> flag;                                      // device: bit 0 = a pivot vanished / was not finite
  std::vector<BcrLevelHost> levels;
  BcrRange top_inv, top_invupd, top_task;
  int top_m = 0, top_ld = 0;
  int64_t bytes32 = 0, bytes64 = 0, setup_flops = 0;
  int launches_per_solve = 0;
};

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
// dense blocks from the coarse level's 3x3 block-CSR values; a node's own (diagonal) block - marked by a negative leading
// dimension - is scaled by 1 + shift: the level solved is A_c + shift * blockdiag(A_c) (see bcr_refresh)
__global__ void k_bcr_fill(int64_t nblk, const float* __restrict__ cvals, const int64_t* __restrict__ dst,
                           const int32_t* __restrict__ ld, double shift, double* __restrict__ arena) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nblk; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = dst[e];
    if (d < 0) continue;
    const int64_t l = ld[e] < 0 ? -ld[e] : ld[e];
    const double f = ld[e] < 0 ? 1.0 + shift : 1.0;
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < 3; ++j) arena[d + c * l + j] = f * (double)cvals[9 * e + 3 * c + j];
  }
}
__global__ void k_bcr_gather(int64_t nc, const int32_t* __restrict__ pos, const float* __restrict__ rc4, double* __restrict__ b) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nc; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(rc4)[i];
    const int64_t p = 3 * (int64_t)pos[i];
    b[p] = v.x; b[p + 1] = v.y; b[p + 2] = v.z;
  }
}
__global__ void k_bcr_scatter(int64_t nc, const int32_t* __restrict__ pos, const double* __restrict__ x, float* __restrict__ xc4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nc; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = 3 * (int64_t)pos[i];
    reinterpret_cast<float4*>(xc4)[i] = make_float4((float)x[p], (float)x[p + 1], (float)x[p + 2], 0.f);
  }
}

// In-place inverse of m x m blocks (row-major, leading dimension ld) by a blocked Gauss-Jordan without pivoting, 32 columns per
// panel, TWO launches per panel for all blocks of a reduction level at once:
//   k_bcr_panel  (one workgroup per block): invert the 32 x 32 pivot block in LDS; R = Pinv * (row panel), Pinv itself on the
//                panel's own columns; rows of the panel <- R; column panel -> Cb (rows of the panel zeroed), then zeroed in A;
//   k_bcr_gemm   (64 x 64 tiles, matrix pipe):  A += -Cb * R,   i.e.  A_ij <- A_ij - A_ik Pinv A_kj  and  A_ik <- -A_ik Pinv.
// (Round 5's first version did the whole inversion in ONE workgroup per block: 12.6 ms per launch at 432 unknowns, 113 ms per
// refresh at 1.12 M tets - the update of the block was a latency-bound read-modify-write loop of a single compute unit.)
__global__ __launch_bounds__(256) void k_bcr_panel(const BcrInv* __restrict__ tasks, int k0, double* __restrict__ arena,
                                                   int32_t* __restrict__ flag) {
  __shared__ double Ps[BCR_PANEL][BCR_PANEL + 1], Qs[BCR_PANEL][BCR_PANEL + 1];
  const BcrInv T = tasks[blockIdx.x];
  const int m = T.m, ld = T.ld, tid = threadIdx.x;
  double* A = arena + T.a;
  double* Cb = arena + T.cb;
  double* Rb = arena + T.rb;
  if (k0 >= m) {                                   // a smaller block of the batch: nothing left to eliminate, the update adds zero
    for (int idx = tid; idx < m * BCR_PANEL; idx += 256) Cb[idx] = 0.0;
    return;
  }
  const int nbk = min(BCR_PANEL, m - k0);
  for (int idx = tid; idx < BCR_PANEL * BCR_PANEL; idx += 256) {
    const int r = idx / BCR_PANEL, c = idx - r * BCR_PANEL;
    Ps[r][c] = (r < nbk && c < nbk) ? A[(size_t)(k0 + r) * ld + k0 + c] : (r == c ? 1.0 : 0.0);
    Qs[r][c] = r == c ? 1.0 : 0.0;
  }
  __syncthreads();
  // Gauss-Jordan on [P | Q]: after pivot p the columns <= p of P are unit vectors and are not touched again, so the factors
  // P[r][p] are read from a column nobody writes - two barriers per pivot
  for (int p = 0; p < nbk; ++p) {
    const double piv = Ps[p][p];
    const bool bad = !(fabs(piv) > 1e-290) || !isfinite(piv);
    if (bad && tid == 0) atomicOr(flag, 1);
    const double ip = bad ? 1.0 : 1.0 / piv;
    if (tid < 2 * BCR_PANEL) {
      const int c = tid & (BCR_PANEL - 1);
      if (tid < BCR_PANEL) { if (c > p) Ps[p][c] *= ip; }
      else Qs[p][c] *= ip;
    }
    __syncthreads();
    for (int idx = tid; idx < BCR_PANEL * 2 * BCR_PANEL; idx += 256) {
      const int r = idx / (2 * BCR_PANEL), cc = idx - r * 2 * BCR_PANEL;
      if (r == p || r >= nbk) continue;
      const double f = Ps[r][p];
      if (cc < BCR_PANEL) { if (cc > p) Ps[r][cc] -= f * Ps[p][cc]; }
      else Qs[r][cc - BCR_PANEL] -= f * Qs[p][cc - BCR_PANEL];
    }
    __syncthreads();
  }
  // R = Pinv * A[panel rows, :]  (Pinv on the panel's own columns) -> Rb; Cb = A[:, panel columns] with the panel's rows zeroed
  for (int j = tid; j < m; j += 256) {
    double col[BCR_PANEL];
    for (int s = 0; s < BCR_PANEL; ++s) col[s] = s < nbk ? A[(size_t)(k0 + s) * ld + j] : 0.0;
    const bool inpanel = j >= k0 && j < k0 + nbk;
    for (int r = 0; r < BCR_PANEL; ++r) {
      double v = 0.0;
      if (r < nbk) {
        if (inpanel) v = Qs[r][j - k0];
        else for (int s = 0; s < BCR_PANEL; ++s) v += Qs[r][s] * col[s];
      }
      Rb[(size_t)r * m + j] = v;
    }
  }
  for (int idx = tid; idx < m * BCR_PANEL; idx += 256) {
    const int i = idx / BCR_PANEL, s = idx - i * BCR_PANEL;
    Cb[idx] = (s < nbk && !(i >= k0 && i < k0 + nbk)) ? A[(size_t)i * ld + k0 + s] : 0.0;
  }
  __syncthreads();                                 // every read of the old panels is done
  for (int idx = tid; idx < m * nbk; idx += 256) {
    const int i = idx / nbk, s = idx - i * nbk;
    if (!(i >= k0 && i < k0 + nbk)) A[(size_t)i * ld + k0 + s] = 0.0;
  }
  for (int j = tid; j < m; j += 256)
    for (int r = 0; r < nbk; ++r) A[(size_t)(k0 + r) * ld + j] = Rb[(size_t)r * m + j];
}
// the finished inverses as FP32 operators of the solve (and the finiteness check of the whole set-up)
__global__ __launch_bounds__(256) void k_bcr_copy32(const BcrInv* __restrict__ tasks, const double* __restrict__ arena,
                                                    float* __restrict__ arena32, int32_t* __restrict__ flag) {
  const BcrInv T = tasks[blockIdx.x];
  if (T.o32 < 0) return;
  const double* A = arena + T.a;
  float* O = arena32 + T.o32;
  bool bad = false;
  for (int idx = threadIdx.x; idx < T.m * T.m; idx += 256) {
    const int i = idx / T.m, j = idx - i * T.m;
    const double v = A[(size_t)i * T.ld + j];
    bad = bad || !isfinite(v);
    O[(size_t)i * T.ld32 + j] = (float)v;
  }
  if (bad) atomicOr(flag, 1);
}

// Batched C = beta C + alpha (A1 B1 + A2 B2) on the FP64 matrix pipe.  One workgroup (four waves, 2 x 2) per 64 x 64 tile of
// C, a wave owns 32 x 32 = 2 x 2 tiles of v_mfma_f64_16x16x4_f64 (A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane
// & 15], D[row = (lane >> 4) + 4 reg][col = lane & 15]: tools/mfma_layout_check.hip).  Operands come straight from L1 / L2:
// the A fragment of a wave is 16 rows x 32 bytes, the B fragment four 128-byte runs; the blocks are a few MB and every tile
// row / column is re-read by the 4 - 7 workgroups beside it.
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bcr_gemm(const BcrGemmTile* __restrict__ tiles, const BcrGemm* __restrict__ tasks,
                                                  double* __restrict__ arena, float* __restrict__ arena32) {
  const BcrGemmTile tl = tiles[blockIdx.x];
  const BcrGemm G = tasks[tl.task];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = 64 * tl.ti + 32 * (wave >> 1), j0 = 64 * tl.tj + 32 * (wave & 1);
  const int lm = lane & 15, lk = lane >> 4;
  v4d acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  for (int prod = 0; prod < 2; ++prod) {
    const int64_t ao = prod ? G.a2 : G.a1, bo = prod ? G.b2 : G.b1;
    const int K = prod ? G.K2 : G.K1, lda = prod ? G.lda2 : G.lda1, ldb = prod ? G.ldb2 : G.ldb1;
    if (ao < 0 || bo < 0 || K <= 0) continue;
    const double* A = arena + ao;
    const double* B = arena + bo;
    for (int k0 = 0; k0 < K; k0 += 4) {
      const int k = k0 + lk;
      double af[2], bf[2];
      for (int t = 0; t < 2; ++t) {
        const int row = i0 + 16 * t + lm, col = j0 + 16 * t + lm;
        af[t] = (row < G.M && k < K) ? A[(size_t)row * lda + k] : 0.0;
        bf[t] = (col < G.N && k < K) ? B[(size_t)k * ldb + col] : 0.0;
      }
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
  double* C = G.c >= 0 ? arena + G.c : nullptr;
  float* O = G.o32 >= 0 ? arena32 + G.o32 : nullptr;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + 16 * a + lk + 4 * r, col = j0 + 16 * b + lm;
        if (row >= G.M || col >= G.N) continue;
        double v = G.alpha * acc[a][b][r];
        if (C) {
          if (G.beta != 0.0) v += G.beta * C[(size_t)row * G.ldc + col];
          C[(size_t)row * G.ldc + col] = v;
        }
        if (O) O[(size_t)row * G.ld32 + col] = (float)v;
      }
}

// One launch of the solve: every workgroup takes 16 rows of one task, stages the task's input vector (<= three segments of b /
// x) in LDS and streams its rows of W (FP32, float4 per lane) against it; a wave per row, four rows per wave.
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_bcr_apply(const BcrTile* __restrict__ tiles, const BcrTask* __restrict__ tasks,
                                                   const float* __restrict__ W, double* __restrict__ b, double* __restrict__ x) {
  extern __shared__ double in[];
  const BcrTile tl = tiles[blockIdx.x];
  const BcrTask T = tasks[tl.task];
  int c0 = 0;
  for (int s = 0; s < T.nseg; ++s) {
    const double* src = (T.seg[s].src ? x : b) + T.seg[s].off;
    for (int i = threadIdx.x; i < T.seg[s].len; i += 256) in[c0 + i] = src[i];
    c0 += T.seg[s].len;
  }
  for (int i = c0 + threadIdx.x; i < T.ldw; i += 256) in[i] = 0.0;
  __syncthreads();
  // a wave takes four consecutive rows and keeps their loads in flight together (one row at a time was a chain of four
  // memory latencies per workgroup: 21 us per launch at 1.12 M tets for 5 us of bytes)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = tl.row0 + 4 * wave;
  if (r0 >= T.rows) return;
  const float* wr[4];
  for (int k = 0; k < 4; ++k) wr[k] = W + T.w + (size_t)min(r0 + k, T.rows - 1) * T.ldw;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int c = 4 * lane; c < T.ldw; c += 256) {
    float4 w[4];
    for (int k = 0; k < 4; ++k) w[k] = *reinterpret_cast<const float4*>(wr[k] + c);
    const double i0 = in[c], i1 = in[c + 1], i2 = in[c + 2], i3 = in[c + 3];
    for (int k = 0; k < 4; ++k) acc[k] += (double)w[k].x * i0 + (double)w[k].y * i1 + (double)w[k].z * i2 + (double)w[k].w * i3;
  }
  for (int k = 0; k < 4; ++k) {
    const double v = wave_sum_dpp(acc[k]);
    if (lane == 0 && r0 + k < T.rows) {
      if (FORWARD) b[T.out + r0 + k] += v;
      else x[T.out + r0 + k] = v;
    }
  }
}

namespace host {

static int64_t grid1(int64_t n) { return std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

void bcr_free(FsiCtx* ctx) {
  if (!ctx->bcr) return;
  BcrData* d = ctx->bcr;
  d->pos.release(); d->fill_dst.release(); d->fill_ld.release(); d->arena64.release(); d->b.release(); d->x.release();
  d->arena32.release(); d->tasks.release(); d->tiles.release(); d->gemms.release(); d->gtiles.release(); d->invs.release();
  d->flag.release();
  delete d;
  ctx->bcr = nullptr;
}

// Host planning: BFS levels of the coarse vertex graph from one end of the wall, the reduction schedule and every descriptor
// table.  `cptr / ccol`: the coarse level's block-CSR pattern (nc nodes).  Returns FSI_OK with ctx->bcr == nullptr when the
// level does not suit the method (a BFS level wider than `max_m` unknowns, operators above `max_bytes`, an edge across more
// than one level): the sweeps stay.
int bcr_plan(FsiCtx* ctx, int64_t nc, const std::vector<int64_t>& cptr, const std::vector<int32_t>& ccol, BcrPlanStats* stats) {
  if (ctx) bcr_free(ctx);
  const int max_m = 2000;
  const double max_bytes = 3.0e9;
  if (nc <= 0) return FSI_OK;
  // ---- BFS levels: from an arbitrary node to its far end, then from that whole end set (level sets = cross-sections) ----
  std::vector<int32_t> level(nc, -1);
  auto bfs = [&](const std::vector<int32_t>& seeds, std::vector<int32_t>& lv) {
    std::vector<int32_t> front = seeds, next;
    for (int32_t s : seeds) lv[s] = 0;
    int32_t L = 0;
    while (!front.empty()) {
      next.clear();
      for (int32_t u : front)
        for (int64_t e = cptr[u]; e < cptr[u + 1]; ++e) {
          const int32_t v = ccol[e];
          if (lv[v] < 0) { lv[v] = L + 1; next.push_back(v); }
        }
      front.swap(next);
      ++L;
    }
    return L;      // number of levels reached
  };
  int32_t nlev = 0;
  {
    std::vector<int32_t> tmp(nc, -1);
    for (int64_t root = 0; root < nc; ++root) {          // component by component; levels of all components share the blocks
      if (level[root] >= 0) continue;
      std::fill(tmp.begin(), tmp.end(), -1);
      const int32_t L1 = bfs({(int32_t)root}, tmp);
      std::vector<int32_t> far;
      for (int64_t i = 0; i < nc; ++i) if (tmp[i] == L1 - 1) far.push_back((int32_t)i);
      const int32_t L2 = bfs(far, level);
      nlev = std::max(nlev, L2);
    }
  }
  const int64_t K = nlev;
  std::vector<int64_t> lcount(K + 1, 0);
  for (int64_t i = 0; i < nc; ++i) lcount[level[i] + 1] += 1;
  for (int64_t k = 0; k < K; ++k) lcount[k + 1] += lcount[k];
  std::vector<int32_t> pos(nc);
  {
    std::vector<int64_t> fillp(lcount.begin(), lcount.end() - 1);
    for (int64_t i = 0; i < nc; ++i) pos[i] = (int32_t)(fillp[level[i]]++);       // ascending node id inside a level
  }
  std::vector<int32_t> msz(K), moff(K);
  int max_block = 0;
  for (int64_t k = 0; k < K; ++k) { msz[k] = (int32_t)(3 * (lcount[k + 1] - lcount[k])); moff[k] = (int32_t)(3 * lcount[k]); max_block = std::max(max_block, (int)msz[k]); }
  if (stats) { stats->blocks = K; stats->max_block = max_block; stats->usable = 0; }
  if (max_block > max_m) return FSI_OK;
  for (int64_t i = 0; i < nc; ++i)
    for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e)
      if (std::abs(level[ccol[e]] - level[i]) > 1) return FSI_OK;              // (an asymmetric pattern: not block tridiagonal)

  BcrData* d = new BcrData();
  d->nc = nc; d->n = 3 * nc; d->K = K; d->max_block = max_block;
  // ---- FP64 arena: level 0 = D_k, L_k, U_k of every block, then the buffers of the reduction levels --------------------------
  int64_t a64 = 0, a32 = 0;
  auto alloc64 = [&](int64_t rows, int64_t cols) { const int64_t o = a64; a64 += rows * cols; return o; };
  auto ld4 = [](int c) { return (c + 3) & ~3; };
  auto alloc32 = [&](int64_t rows, int ld) { const int64_t o = a32; a32 += rows * (int64_t)ld; a32 = (a32 + 3) & ~(int64_t)3; return o; };
  struct Blk { int64_t D, L, U; int lcols, ucols; };            // L couples to the left ACTIVE neighbour (lcols columns), U to the right
  std::vector<Blk> blk(K);
  for (int64_t k = 0; k < K; ++k) {
    blk[k].D = alloc64(msz[k], msz[k]);
    blk[k].lcols = k > 0 ? msz[k - 1] : 0;
    blk[k].ucols = k + 1 < K ? msz[k + 1] : 0;
    blk[k].L = k > 0 ? alloc64(msz[k], msz[k - 1]) : -1;
    blk[k].U = k + 1 < K ? alloc64(msz[k], msz[k + 1]) : -1;
  }
  d->level0_doubles = a64;
  // sparse -> dense destinations
  std::vector<int64_t> fdst(ccol.size(), -1);
  std::vector<int32_t> fld(ccol.size(), 0);
  for (int64_t i = 0; i < nc; ++i)
    for (int64_t e = cptr[i]; e < cptr[i + 1]; ++e) {
      const int32_t j = ccol[e];
      const int ki = level[i], kj = level[j];
      const int64_t pi = 3 * (int64_t)pos[i] - moff[ki], pj = 3 * (int64_t)pos[j] - moff[kj];
      int64_t base; int ld;
      if (kj == ki) { base = blk[ki].D; ld = msz[ki]; }
      else if (kj == ki - 1) { base = blk[ki].L; ld = msz[kj]; }
      else { base = blk[ki].U; ld = msz[kj]; }
      fdst[e] = base + pi * ld + pj;
      fld[e] = j == i ? -ld : ld;
    }
  // ---- the reduction schedule ------------------------------------------------------------------------------------------
  std::vector<BcrTask> tasks;
  std::vector<BcrTile> tiles;
  std::vector<BcrGemm> gemms;
  std::vector<BcrGemmTile> gtiles;
  std::vector<BcrInv> invs;
  auto add_gemm = [&](const BcrGemm& g) {
    const int32_t id = (int32_t)gemms.size();
    gemms.push_back(g);
    for (int ti = 0; ti < (g.M + 63) / 64; ++ti)
      for (int tj = 0; tj < (g.N + 63) / 64; ++tj) gtiles.push_back(BcrGemmTile{id, ti, tj});
    d->setup_flops += 2LL * g.M * g.N * ((int64_t)g.K1 + g.K2);
  };
  // an in-place inverse: its panel scratch and the tiles of its rank-32 updates A += -Cb Rb (one k_bcr_gemm launch per panel)
  auto add_inverse = [&](int64_t D, int64_t o32, int m, int ld32) {
    const int64_t cb = alloc64(m, BCR_PANEL), rb = alloc64(BCR_PANEL, m);
    invs.push_back(BcrInv{D, o32, cb, rb, m, m, ld32});
    add_gemm(BcrGemm{cb, rb, -1, -1, D, -1, m, m, BCR_PANEL, 0, BCR_PANEL, m, 0, 0, m, 0, -1.0, 1.0});
    d->setup_flops += 2LL * m * m * (int64_t)(m - BCR_PANEL);       // (add_gemm counted one panel; the update runs once per panel)
  };
  auto add_task = [&](const BcrTask& t) {
    const int32_t id = (int32_t)tasks.size();
    tasks.push_back(t);
    for (int r0 = 0; r0 < t.rows; r0 += 16) tiles.push_back(BcrTile{id, r0});
  };
  std::vector<int32_t> active(K);
  std::iota(active.begin(), active.end(), 0);
  while (active.size() > 1) {
    BcrLevelHost lv;
    const size_t na = active.size();
    // (1) inverses of the eliminated blocks, FP32 copy straight into the backward operator
    lv.inv.first = (int64_t)invs.size();
    lv.invupd.first = (int64_t)gtiles.size();
    std::vector<int64_t> wb(na, -1);            // backward operator of the eliminated block at position i
    std::vector<int> wb_ld(na, 0);
    for (size_t i = 1; i < na; i += 2) {
      const int e = active[i], a = active[i - 1], c = i + 1 < na ? active[i + 1] : -1;
      const int cols = msz[e] + msz[a] + (c >= 0 ? msz[c] : 0);
      wb_ld[i] = ld4(cols);
      wb[i] = alloc32(msz[e], wb_ld[i]);
      add_inverse(blk[e].D, wb[i], msz[e], wb_ld[i]);
      lv.inv_maxm = std::max(lv.inv_maxm, (int)msz[e]);
    }
    lv.inv.count = (int64_t)invs.size() - lv.inv.first;
    lv.invupd.count = (int64_t)gtiles.size() - lv.invupd.first;
    // (2) first batch of products: H_ea = -Dinv_e L_e, H_ec = -Dinv_e U_e (FP32 only), G_jl = -L_j Dinv_l, G_jr = -U_j Dinv_r
    lv.gemm1.first = (int64_t)gtiles.size();
    for (size_t i = 1; i < na; i += 2) {
      const int e = active[i], a = active[i - 1], c = i + 1 < na ? active[i + 1] : -1;
      add_gemm(BcrGemm{blk[e].D, blk[e].L, -1, -1, -1, wb[i] + msz[e], msz[e], msz[a], msz[e], 0, msz[e], blk[e].lcols, 0, 0, 0, wb_ld[i], -1.0, 0.0});
      if (c >= 0)
        add_gemm(BcrGemm{blk[e].D, blk[e].U, -1, -1, -1, wb[i] + msz[e] + msz[a], msz[e], msz[c], msz[e], 0, msz[e], blk[e].ucols, 0, 0, 0, wb_ld[i], -1.0, 0.0});
    }
    std::vector<int64_t> gl(na, -1), gr(na, -1), wf(na, -1);
    std::vector<int> wf_ld(na, 0);
    for (size_t i = 0; i < na; i += 2) {
      const int j = active[i], l = i > 0 ? active[i - 1] : -1, r = i + 1 < na ? active[i + 1] : -1;
      const int cols = (l >= 0 ? msz[l] : 0) + (r >= 0 ? msz[r] : 0);
      if (cols == 0) continue;
      wf_ld[i] = ld4(cols);
      wf[i] = alloc32(msz[j], wf_ld[i]);
      if (l >= 0) {
        gl[i] = alloc64(msz[j], msz[l]);
        add_gemm(BcrGemm{blk[j].L, blk[l].D, -1, -1, gl[i], wf[i], msz[j], msz[l], msz[l], 0, blk[j].lcols, msz[l], 0, 0, msz[l], wf_ld[i], -1.0, 0.0});
      }
      if (r >= 0) {
        gr[i] = alloc64(msz[j], msz[r]);
        add_gemm(BcrGemm{blk[j].U, blk[r].D, -1, -1, gr[i], wf[i] + (l >= 0 ? msz[l] : 0), msz[j], msz[r], msz[r], 0, blk[j].ucols, msz[r], 0, 0, msz[r], wf_ld[i], -1.0, 0.0});
      }
    }
    lv.gemm1.count = (int64_t)gtiles.size() - lv.gemm1.first;
    // (3) second batch: D_j += G_jl U_l + G_jr L_r ; L_j' = G_jl L_l ; U_j' = G_jr U_r
    lv.gemm2.first = (int64_t)gtiles.size();
    std::vector<Blk> nb_(na);
    for (size_t i = 0; i < na; i += 2) {
      const int j = active[i], l = i > 0 ? active[i - 1] : -1, r = i + 1 < na ? active[i + 1] : -1;
      Blk nbk = blk[j];
      if (l >= 0 || r >= 0) {
        BcrGemm g{-1, -1, -1, -1, blk[j].D, -1, msz[j], msz[j], 0, 0, 0, 0, 0, 0, msz[j], 0, 1.0, 1.0};
        if (l >= 0) { g.a1 = gl[i]; g.b1 = blk[l].U; g.K1 = msz[l]; g.lda1 = msz[l]; g.ldb1 = blk[l].ucols; }
        if (r >= 0) { g.a2 = gr[i]; g.b2 = blk[r].L; g.K2 = msz[r]; g.lda2 = msz[r]; g.ldb2 = blk[r].lcols; }
        add_gemm(g);
      }
      nbk.L = -1; nbk.lcols = 0; nbk.U = -1; nbk.ucols = 0;
      if (l >= 0 && i >= 2) {                     // the left neighbour's own left coupling: to the survivor two positions down
        const int ll = active[i - 2];
        nbk.L = alloc64(msz[j], msz[ll]); nbk.lcols = msz[ll];
        add_gemm(BcrGemm{gl[i], blk[l].L, -1, -1, nbk.L, -1, msz[j], msz[ll], msz[l], 0, msz[l], blk[l].lcols, 0, 0, msz[ll], 0, 1.0, 0.0});
      }
      if (r >= 0 && i + 2 < na) {
        const int rr = active[i + 2];
        nbk.U = alloc64(msz[j], msz[rr]); nbk.ucols = msz[rr];
        add_gemm(BcrGemm{gr[i], blk[r].U, -1, -1, nbk.U, -1, msz[j], msz[rr], msz[r], 0, msz[r], blk[r].ucols, 0, 0, msz[rr], 0, 1.0, 0.0});
      }
      nb_[i] = nbk;
    }
    lv.gemm2.count = (int64_t)gtiles.size() - lv.gemm2.first;
    // solve tasks of this level
    lv.fwd.first = (int64_t)tiles.size();
    for (size_t i = 0; i < na; i += 2) {
      if (wf[i] < 0) continue;
      const int j = active[i], l = i > 0 ? active[i - 1] : -1, r = i + 1 < na ? active[i + 1] : -1;
      BcrTask t{wf[i], msz[j], wf_ld[i], moff[j], 0, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
      if (l >= 0) t.seg[t.nseg++] = BcrSeg{moff[l], msz[l], 0};
      if (r >= 0) t.seg[t.nseg++] = BcrSeg{moff[r], msz[r], 0};
      add_task(t);
      lv.fwd_maxld = std::max(lv.fwd_maxld, wf_ld[i]);
    }
    lv.fwd.count = (int64_t)tiles.size() - lv.fwd.first;
    lv.bwd.first = (int64_t)tiles.size();
    for (size_t i = 1; i < na; i += 2) {
      const int e = active[i], a = active[i - 1], c = i + 1 < na ? active[i + 1] : -1;
      BcrTask t{wb[i], msz[e], wb_ld[i], moff[e], 0, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
      t.seg[t.nseg++] = BcrSeg{moff[e], msz[e], 0};
      t.seg[t.nseg++] = BcrSeg{moff[a], msz[a], 1};
      if (c >= 0) t.seg[t.nseg++] = BcrSeg{moff[c], msz[c], 1};
      add_task(t);
      lv.bwd_maxld = std::max(lv.bwd_maxld, wb_ld[i]);
    }
    lv.bwd.count = (int64_t)tiles.size() - lv.bwd.first;
    d->levels.push_back(lv);
    std::vector<int32_t> surv;
    for (size_t i = 0; i < na; i += 2) { blk[active[i]] = nb_[i]; surv.push_back(active[i]); }
    active.swap(surv);
  }
  {
    const int t = active[0];
    d->top_m = msz[t]; d->top_ld = ld4(msz[t]);
    const int64_t w = alloc32(msz[t], d->top_ld);
    d->top_inv = BcrRange{(int64_t)invs.size(), 1};
    d->top_invupd.first = (int64_t)gtiles.size();
    add_inverse(blk[t].D, w, msz[t], d->top_ld);
    d->top_invupd.count = (int64_t)gtiles.size() - d->top_invupd.first;
    d->top_task = BcrRange{(int64_t)tiles.size(), 0};
    BcrTask tk{w, msz[t], d->top_ld, moff[t], 1, {{moff[t], msz[t], 0}, {0, 0, 0}, {0, 0, 0}}};
    add_task(tk);
    d->top_task.count = (int64_t)tiles.size() - d->top_task.first;
  }
  d->bytes32 = a32 * 4; d->bytes64 = a64 * 8;
  d->launches_per_solve = 2 * (int)d->levels.size() + 1;      // (+ 2 when the vectors are not handed over in the solve's own order)
  if (stats) {
    stats->levels = (int64_t)d->levels.size(); stats->bytes32 = d->bytes32; stats->bytes64 = d->bytes64; stats->setup_flops = d->setup_flops;
    stats->launches = d->launches_per_solve;
  }
  if ((double)d->bytes32 > max_bytes || (double)d->bytes64 > 4.0 * max_bytes) { delete d; return FSI_OK; }
  if (stats) stats->usable = 1;
  if (stats && stats->plan_only) {               // host-side dry run (tests without a device): hand the permutation out, keep nothing
    if (stats->pos_out) std::copy(pos.begin(), pos.end(), stats->pos_out);
    if (stats->level_out) std::copy(level.begin(), level.end(), stats->level_out);
    delete d;
    return FSI_OK;
  }
  ctx->bcr = d;
  FSICHK(upload(ctx, d->pos, pos));
  FSICHK(upload(ctx, d->fill_dst, fdst));
  FSICHK(upload(ctx, d->fill_ld, fld));
  d->nfill = (int64_t)fdst.size();
  FSICHK(upload(ctx, d->tasks, tasks));
  FSICHK(upload(ctx, d->tiles, tiles));
  FSICHK(upload(ctx, d->gemms, gemms));
  FSICHK(upload(ctx, d->gtiles, gtiles));
  FSICHK(upload(ctx, d->invs, invs));
  HIPCHK(d->arena64.alloc((size_t)a64));
  HIPCHK(d->arena32.alloc((size_t)a32 + 4));
  HIPCHK(d->b.alloc((size_t)d->n));
  HIPCHK(d->x.alloc((size_t)d->n));
  HIPCHK(d->flag.alloc(4));
  d->planned = true;
  if (getenv("FSI_DEBUG"))
    fprintf(stderr, "[fsi] solid coarse level by block cyclic reduction: %lld nodes in %lld blocks (largest %d unknowns), %zu reduction levels, "
            "operators %.1f MB (FP32), set-up arena %.1f MB, %.2f Gflop per refresh, %d launches per solve\n", (long long)nc, (long long)K,
            max_block, d->levels.size(), d->bytes32 / 1e6, d->bytes64 / 1e6, d->setup_flops / 1e9, d->launches_per_solve);
  return FSI_OK;
}

// New Jacobian: dense blocks from the coarse level's values (ctx->sbmg_cvals after k_sbmg_coarse_finish), then the operators.
int bcr_refresh(FsiCtx* ctx) {
  BcrData* d = ctx->bcr;
  if (!d || !d->planned) return FSI_OK;
  hipStream_t st = ctx->stream;
  d->ready = false;
  HIPCHK(hipMemsetAsync(d->arena64.p, 0, (size_t)d->level0_doubles * sizeof(double), st));
  HIPCHK(hipMemsetAsync(d->flag.p, 0, 4 * sizeof(int32_t), st));
  // What is solved is (A_c + shift * blockdiag(A_c)) x = r.  The fine level of the cycle sweeps on ROUNDED matrix values (FP16
  // records: 5e-4 of an entry), so the level's operator and the fine operator disagree on modes whose eigenvalue is below that
  // rounding - an exact A_c^-1 amplifies exactly those (a factor 1 - lambda~ / lambda of either sign), where the truncated Chebyshev
  // solve it replaces never inverted anything below lmax / kappa.  The shift is that floor as a Tikhonov term: modes above it are
  // solved exactly, modes below it are damped as before (FsiTuning.bcr_shift; 0 = the exact level).
  hipLaunchKernelGGL(k_bcr_fill, dim3((unsigned)grid1(d->nfill)), dim3(256), 0, st, d->nfill, ctx->sbmg_cvals.p, d->fill_dst.p, d->fill_ld.p,
                     ctx->tune.bcr_shift, d->arena64.p);
  auto invert = [&](const BcrRange& r, const BcrRange& upd, int maxm) -> int {
    if (r.count == 0) return FSI_OK;
    for (int k0 = 0; k0 < maxm; k0 += BCR_PANEL) {
      hipLaunchKernelGGL(k_bcr_panel, dim3((unsigned)r.count), dim3(256), 0, st, d->invs.p + r.first, k0, d->arena64.p, d->flag.p);
      hipLaunchKernelGGL(k_bcr_gemm, dim3((unsigned)upd.count), dim3(256), 0, st, d->gtiles.p + upd.first, d->gemms.p, d->arena64.p, d->arena32.p);
    }
    hipLaunchKernelGGL(k_bcr_copy32, dim3((unsigned)r.count), dim3(256), 0, st, d->invs.p + r.first, d->arena64.p, d->arena32.p, d->flag.p);
    return FSI_OK;
  };
  for (const BcrLevelHost& lv : d->levels) {
    FSICHK(invert(lv.inv, lv.invupd, lv.inv_maxm));
    if (lv.gemm1.count)
      hipLaunchKernelGGL(k_bcr_gemm, dim3((unsigned)lv.gemm1.count), dim3(256), 0, st, d->gtiles.p + lv.gemm1.first, d->gemms.p, d->arena64.p, d->arena32.p);
    if (lv.gemm2.count)
      hipLaunchKernelGGL(k_bcr_gemm, dim3((unsigned)lv.gemm2.count), dim3(256), 0, st, d->gtiles.p + lv.gemm2.first, d->gemms.p, d->arena64.p, d->arena32.p);
  }
  FSICHK(invert(d->top_inv, d->top_invupd, d->top_m));
  HIPCHK(hipGetLastError());
  int32_t flag[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(flag, d->flag.p, sizeof flag, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  d->ready = flag[0] == 0;
  if (!d->ready && getenv("FSI_DEBUG")) fprintf(stderr, "[fsi] block cyclic reduction: a pivot vanished or was not finite - the coarse level keeps its sweeps for this Jacobian\n");
  return FSI_OK;
}

// x_c = A_c^-1 r_c on the float4-padded coarse vectors of the solid cycle (rc4 in, xc4 out), queued on `st`.
int bcr_solve(FsiCtx* ctx, const float* rc4, float* xc4, hipStream_t st) {
  BcrData* d = ctx->bcr;
  if (rc4) hipLaunchKernelGGL(k_bcr_gather, dim3((unsigned)grid1(d->nc)), dim3(256), 0, st, d->nc, d->pos.p, rc4, d->b.p);
  for (const BcrLevelHost& lv : d->levels)
    if (lv.fwd.count)
      hipLaunchKernelGGL(k_bcr_apply<true>, dim3((unsigned)lv.fwd.count), dim3(256), (size_t)lv.fwd_maxld * sizeof(double), st,
                         d->tiles.p + lv.fwd.first, d->tasks.p, d->arena32.p, d->b.p, d->x.p);
  hipLaunchKernelGGL(k_bcr_apply<false>, dim3((unsigned)d->top_task.count), dim3(256), (size_t)d->top_ld * sizeof(double), st,
                     d->tiles.p + d->top_task.first, d->tasks.p, d->arena32.p, d->b.p, d->x.p);
  for (auto it = d->levels.rbegin(); it != d->levels.rend(); ++it)
    if (it->bwd.count)
      hipLaunchKernelGGL(k_bcr_apply<false>, dim3((unsigned)it->bwd.count), dim3(256), (size_t)it->bwd_maxld * sizeof(double), st,
                         d->tiles.p + it->bwd.first, d->tasks.p, d->arena32.p, d->b.p, d->x.p);
  if (xc4) hipLaunchKernelGGL(k_bcr_scatter, dim3((unsigned)grid1(d->nc)), dim3(256), 0, st, d->nc, d->pos.p, d->x.p, xc4);
  return FSI_OK;
}
const int32_t* bcr_pos(const FsiCtx* ctx) { return ctx->bcr->pos.p; }
double* bcr_rhs(FsiCtx* ctx) { return ctx->bcr->b.p; }
const double* bcr_sol(const FsiCtx* ctx) { return ctx->bcr->x.p; }

bool bcr_ready(const FsiCtx* ctx) { return ctx->bcr && ctx->bcr->planned && ctx->bcr->ready; }

}  // namespace host
}  // namespace fsi

// ---- C-ABI: planning dry run (host only) and the test hooks of the coarse solve ---------------------------------------------
using namespace fsi;
using namespace fsi::host;

extern "C" {

int fsi_bcr_plan_graph(int64_t nc, const int64_t* cptr, const int32_t* ccol, int64_t* stats_out, int32_t* pos_out, int32_t* level_out) {
  if (nc <= 0 || !cptr || !ccol || !stats_out) return FSI_ERR_INVALID;
  std::vector<int64_t> p(cptr, cptr + nc + 1);
  std::vector<int32_t> c(ccol, ccol + cptr[nc]);
  for (int32_t v : c) if (v < 0 || v >= nc) return FSI_ERR_INVALID;
  BcrPlanStats st;
  st.plan_only = 1; st.pos_out = pos_out; st.level_out = level_out;
  const int rc = bcr_plan(nullptr, nc, p, c, &st);
  stats_out[0] = st.usable; stats_out[1] = st.blocks; stats_out[2] = st.max_block; stats_out[3] = st.levels;
  stats_out[4] = st.bytes32; stats_out[5] = st.bytes64; stats_out[6] = st.setup_flops; stats_out[7] = st.launches;
  return rc;
}

int fsi_solid_coarse_info(const FsiCtx* ctx, int64_t* out) {
  if (!ctx || !out) return FSI_ERR_INVALID;
  const BcrData* d = ctx->bcr;
  out[0] = ctx->sbmg_nc; out[1] = ctx->sbmg_nblk; out[2] = d && d->planned; out[3] = bcr_ready(ctx);
  out[4] = d ? d->K : 0; out[5] = d ? (int64_t)d->levels.size() : 0; out[6] = d ? d->bytes32 : 0; out[7] = d ? d->launches_per_solve : 0;
  out[8] = d ? d->max_block : 0; out[9] = ctx->bcr_solves; out[10] = d ? d->setup_flops : 0; out[11] = ctx->sbmg_ready;
  return FSI_OK;
}

int fsi_solid_coarse_matrix(FsiCtx* ctx, int64_t* cptr, int32_t* ccol, float* cvals) {
  if (!ctx || !cptr || !ccol || !cvals || ctx->sbmg_nc <= 0) return FSI_ERR_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cptr, ctx->sbmg_cptr.p, (size_t)(ctx->sbmg_nc + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(ccol, ctx->sbmg_ccol.p, (size_t)ctx->sbmg_nblk * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cvals, ctx->sbmg_cvals.p, (size_t)9 * ctx->sbmg_nblk * sizeof(float), hipMemcpyDeviceToHost));
  return FSI_OK;
}

// x = A_c^-1 rhs through the production kernels (rhs, x: 3 doubles per coarse node in the level's own node order)
int fsi_solid_coarse_solve(FsiCtx* ctx, const double* rhs, double* x) {
  if (!ctx || !rhs || !x) return FSI_ERR_INVALID;
  if (!bcr_ready(ctx)) { ctx->err = "the exact coarse solve is not available on this context / Jacobian"; return FSI_ERR_INVALID; }
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t nc = ctx->sbmg_nc;
  std::vector<float> r4(4 * (size_t)nc, 0.f), x4(4 * (size_t)nc, 0.f);
  for (int64_t i = 0; i < nc; ++i) for (int c = 0; c < 3; ++c) r4[4 * i + c] = (float)rhs[3 * i + c];
  float* w = ctx->sbmg_work.p;                       // [5][4 nc] floats: the cycle's own coarse work area
  HIPCHK(hipMemcpyAsync(w + 4 * 4 * nc, r4.data(), r4.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  FSICHK(bcr_solve(ctx, w + 4 * 4 * nc, w + 3 * 4 * nc, ctx->stream));
  HIPCHK(hipMemcpyAsync(x4.data(), w + 3 * 4 * nc, x4.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int64_t i = 0; i < nc; ++i) for (int c = 0; c < 3; ++c) x[3 * i + c] = x4[4 * i + c];
  return FSI_OK;
}

}  // extern "C"
