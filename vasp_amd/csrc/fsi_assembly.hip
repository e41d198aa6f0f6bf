// Element-loop kernels of the monolithic ALE-FSI step (gfx950): geometry, residual, Jacobian, boundary terms.
//
// Replaces `assemble(-F)` and `assemble(J_nonlinear)` / `assemble(J_linear)` of turtleFSI's newtonsolver as VaSP
// drives it.  One wavefront (64 lanes) owns one tetrahedron:
//   residual : lanes 0..23 evaluate the 24 quadrature points (interpolation + pointwise flux), results go through
//              LDS, then all 64 lanes contract against the P2/P1 test tables (one local dof per lane) and
//              scatter-add into the global vector;
//   Jacobian : lane j carries the forward-mode derivative along local trial dof j (fsi::Dual), accumulates its
//              64-entry column in registers and scatter-adds it into the CSR values through the per-element
//              neighbour-index tables (no search in the hot loop).
// HBM traffic per tet (algorithmic): residual 10*4 + 2*64*8 + 80 (geometry) + 64*8 write; Jacobian + 64*64*8 write.
#include "fsi_kernels.hpp"

namespace fsi {

__constant__ double c_qw[NQ];
__constant__ double c_N[NQ][10];
__constant__ double c_dN[NQ][10][3];
__constant__ double c_L[NQ][4];

static const int H_TET_EDGES[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};

// FIAT default degree-6 scheme on the tetrahedron: Keast 24 points.
hipError_t upload_tables() {
  double qp[NQ][3], qw[NQ];
  int n = 0;
  const double a4[3] = {0.214602871259151684, 0.040673958534611353, 0.322337890142275646};
  const double w4[3] = {0.039922750258167949, 0.010077211055320643, 0.055357181543654720};
  for (int o = 0; o < 3; ++o) {
    double a = a4[o], b = 1.0 - 3.0 * a;
    double pts[4][3] = {{b, a, a}, {a, a, a}, {a, a, b}, {a, b, a}};
    for (int i = 0; i < 4; ++i) {
      for (int k = 0; k < 3; ++k) qp[n][k] = pts[i][k];
      qw[n++] = w4[o] / 6.0;
    }
  }
  {
    double a = 0.063661001875017525, b = 0.269672331458315867, c = 0.603005664791649076;
    double pts[12][3] = {{b, a, a}, {a, b, a}, {a, a, b}, {c, a, a}, {a, c, a}, {a, a, c},
                         {a, b, c}, {b, c, a}, {c, a, b}, {a, c, b}, {b, a, c}, {c, b, a}};
    for (int i = 0; i < 12; ++i) {
      for (int k = 0; k < 3; ++k) qp[n][k] = pts[i][k];
      qw[n++] = 0.048214285714285714 / 6.0;
    }
  }
  double N[NQ][10], dN[NQ][10][3], L[NQ][4];
  const double dL[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int q = 0; q < NQ; ++q) {
    double l[4] = {1.0 - qp[q][0] - qp[q][1] - qp[q][2], qp[q][0], qp[q][1], qp[q][2]};
    for (int a = 0; a < 4; ++a) {
      L[q][a] = l[a];
      N[q][a] = l[a] * (2.0 * l[a] - 1.0);
      for (int k = 0; k < 3; ++k) dN[q][a][k] = (4.0 * l[a] - 1.0) * dL[a][k];
    }
    for (int e = 0; e < 6; ++e) {
      int i = H_TET_EDGES[e][0], j = H_TET_EDGES[e][1];
      N[q][4 + e] = 4.0 * l[i] * l[j];
      for (int k = 0; k < 3; ++k) dN[q][4 + e][k] = 4.0 * (l[i] * dL[j][k] + l[j] * dL[i][k]);
    }
  }
  hipError_t e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_qw), qw, sizeof(qw))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_N), N, sizeof(N))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_dN), dN, sizeof(dN))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_L), L, sizeof(L))) != hipSuccess) return e;
  return upload_post_tables(qw, &dN[0][0][0], &L[0][0]);       // the same tables for the kernels of fsi_post.hip
}

// ---------------------------------------------------------------------------------------------------------
// geometry: per cell Jinv[k][j] = d xi_k / d x_j (9 doubles) and |det| (1 double)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_geometry(int64_t C, const double* __restrict__ coords, const int32_t* __restrict__ tet_vertices,
                           double* __restrict__ geom) {
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int32_t* t = tet_vertices + 4 * c;
  double x0[3], J[3][3];
  for (int i = 0; i < 3; ++i) x0[i] = coords[3 * (int64_t)t[0] + i];
  for (int k = 0; k < 3; ++k)
    for (int i = 0; i < 3; ++i) J[i][k] = coords[3 * (int64_t)t[k + 1] + i] - x0[i];   // dx_i / dxi_k
  double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  double c01 = J[0][2] * J[2][1] - J[0][1] * J[2][2];
  double c02 = J[0][1] * J[1][2] - J[0][2] * J[1][1];
  double c10 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  double c11 = J[0][0] * J[2][2] - J[0][2] * J[2][0];
  double c12 = J[0][2] * J[1][0] - J[0][0] * J[1][2];
  double c20 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  double c21 = J[0][1] * J[2][0] - J[0][0] * J[2][1];
  double c22 = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  double det = J[0][0] * c00 + J[0][1] * c10 + J[0][2] * c20;
  double r = 1.0 / det;
  double* g = geom + 10 * c;
  g[0] = c00 * r; g[1] = c01 * r; g[2] = c02 * r;
  g[3] = c10 * r; g[4] = c11 * r; g[5] = c12 * r;
  g[6] = c20 * r; g[7] = c21 * r; g[8] = c22 * r;
  g[9] = fabs(det);
}

__device__ inline void load_props(const ElemParams& ep, int kind, int region, FluidProps& fp, SolidProps& sp) {
  if (kind == 0) fp = ep.fluid[region];
  else sp = ep.solid[region];
}

// Interpolate the local dofs (LDS, layout [d_x(10) d_y d_z v_x v_y v_z p(4)]) at quadrature point q.
__device__ inline void interpolate(const double* __restrict__ sU, const double* __restrict__ Jinv, int q,
                                   Kin<double>& s) {
  for (int i = 0; i < 3; ++i) {
    s.d[i] = 0.0; s.v[i] = 0.0;
    for (int j = 0; j < 3; ++j) { s.gd[i][j] = 0.0; s.gv[i][j] = 0.0; }
  }
  s.p = 0.0;
  for (int a = 0; a < 10; ++a) {
    const double N = c_N[q][a];
    const double r0 = c_dN[q][a][0], r1 = c_dN[q][a][1], r2 = c_dN[q][a][2];
    double G[3];
    for (int j = 0; j < 3; ++j) G[j] = r0 * Jinv[j] + r1 * Jinv[3 + j] + r2 * Jinv[6 + j];
    for (int i = 0; i < 3; ++i) {
      const double dv = sU[i * 10 + a], vv = sU[30 + i * 10 + a];
      s.d[i] += N * dv;
      s.v[i] += N * vv;
      for (int j = 0; j < 3; ++j) { s.gd[i][j] += dv * G[j]; s.gv[i][j] += vv * G[j]; }
    }
  }
  for (int a = 0; a < 4; ++a) s.p += c_L[q][a] * sU[60 + a];
}

// Both states (n and n-1) in one pass over the ten nodes: a node's table entries and physical gradient serve the two
// interpolations and are dead afterwards - with two separate passes the compiler keeps the 40 table values of a point in
// registers across both (k_residual then needs 302 VGPRs: one wave per SIMD).  sT[q][a] = (N, dN/dxi_0..2) in LDS.
__device__ inline void interpolate2(const double* __restrict__ sU, const double* __restrict__ sO,
                                    const double* __restrict__ Jinv, const double4* __restrict__ sTq, const double* __restrict__ Lq,
                                    Kin<double>& s, Kin<double>& o) {
  for (int i = 0; i < 3; ++i) {
    s.d[i] = 0.0; s.v[i] = 0.0; o.d[i] = 0.0; o.v[i] = 0.0;
    for (int j = 0; j < 3; ++j) { s.gd[i][j] = 0.0; s.gv[i][j] = 0.0; o.gd[i][j] = 0.0; o.gv[i][j] = 0.0; }
  }
  s.p = 0.0; o.p = 0.0;
#pragma unroll 1
  for (int a = 0; a < 10; ++a) {
    const double4 t = sTq[a];
    const double N = t.x;
    double G[3];
    for (int j = 0; j < 3; ++j) G[j] = t.y * Jinv[j] + t.z * Jinv[3 + j] + t.w * Jinv[6 + j];
    for (int i = 0; i < 3; ++i) {
      const double dv = sU[i * 10 + a], vv = sU[30 + i * 10 + a], dw = sO[i * 10 + a], vw = sO[30 + i * 10 + a];
      s.d[i] += N * dv; s.v[i] += N * vv;
      o.d[i] += N * dw; o.v[i] += N * vw;
      for (int j = 0; j < 3; ++j) {
        s.gd[i][j] += dv * G[j]; s.gv[i][j] += vv * G[j];
        o.gd[i][j] += dw * G[j]; o.gv[i][j] += vw * G[j];
      }
    }
  }
  for (int a = 0; a < 4; ++a) { s.p += Lq[a] * sU[60 + a]; o.p += Lq[a] * sO[60 + a]; }
}

// ---------------------------------------------------------------------------------------------------------
// residual: F += sum over cells of the element vector (un-negated, no boundary conditions)
// ---------------------------------------------------------------------------------------------------------
// Two cells per 64-lane workgroup and round: the quadrature phase (24 points, by far the longest part: two interpolations
// and the flux per point) keeps lanes 0-23 on the first cell and lanes 32-55 on the second, so one instruction stream
// serves 48 lanes instead of 24; the load and the contraction phases take the two cells one after the other with all 64
// lanes.  Workgroups are persistent (a grid-stride loop over the cell pairs) and hold the basis tables in LDS: read from
// constant memory with a lane-dependent index they are vector loads, 96 per cell in the contraction alone, and each costs
// address-unit cycles that an LDS read does not.
// Re != nullptr (the default, FSI_ASSEMBLY unset): the element vector of cell c is STORED to Re[c][64] - a node's six
// entries side by side, [node a][d_x d_y d_z v_x v_y v_z], then the four pressure entries - and k_residual_gather adds, per
// dof, the entries of the cells around the node in ascending cell order: the segmented reduction behind the scatter-add of an
// assembly, done on the owner's side, with sums that are bitwise the same in every run.  Re == nullptr (FSI_ASSEMBLY=atomic):
// scatter-add into F with atomics whose order varies from run to run.
template <int WAVES>      // waves per SIMD the register budget is set for (2: no spills; 3: 56 VGPRs spilled, measured slower)
__global__ __launch_bounds__(64, WAVES) void k_residual(ElemArrays ea, ElemParams ep, const double* __restrict__ U,
                                                        const double* __restrict__ U1, double* __restrict__ F, int64_t C,
                                                        double* __restrict__ Re) {
  const int lane = threadIdx.x;
  __shared__ __attribute__((aligned(32))) double4 sT[NQ][10];        // (N, dN/dxi) per point and node
  __shared__ double sL[NQ][4], sW[NQ];
  __shared__ double sU[2][NLOC], sU1[2][NLOC], sJ[2][10];
  __shared__ double sS[2][NQ][25];
  for (int t = lane; t < NQ * 10; t += 64) {
    const int q = t / 10, a = t % 10;
    sT[q][a] = make_double4(c_N[q][a], c_dN[q][a][0], c_dN[q][a][1], c_dN[q][a][2]);
  }
  for (int t = lane; t < NQ * 4; t += 64) sL[t / 4][t % 4] = c_L[t / 4][t % 4];
  if (lane < NQ) sW[lane] = c_qw[lane];
  const int64_t npairs = (C + 1) / 2;
  const int half = lane >> 5, q = lane & 31;
  // cell pairs in the XCD-aware order of fsi_kernels.hpp: the state entries neighbouring cells share are fetched into ONE L2 instead
  // of eight (gridDim.x is a multiple of 8: the XCD of a logical workgroup is that of its launch index)
  const int64_t span = xcd_span(npairs);
  for (int64_t lw = blockIdx.x; lw < span; lw += gridDim.x) {
    const int64_t pair = xcd_unit(lw, npairs);
    if (pair < 0) continue;
    const int64_t c0 = 2 * pair;
    const int ncell = c0 + 1 < C ? 2 : 1;
    int32_t dof[2] = {0, 0};
    __syncthreads();                                 // the previous pair's contraction has finished with sS / sU
    for (int t = 0; t < ncell; ++t) {
      // the 64 local dofs from 10 node ranks + 4 pressure rows (56 bytes per cell; the 64-entry dof map is 256): lane l < 60
      // is (field l / 30, component (l % 30) / 10, node l % 10) -> dof 6 rank + 3 field + component
      dof[t] = lane < 60 ? 6 * ea.cell_rank[(c0 + t) * 10 + lane % 10] + 3 * (lane / 30) + (lane % 30) / 10
                         : ea.cell_prow[(c0 + t) * 4 + lane - 60];
      sU[t][lane] = U[dof[t]];
      sU1[t][lane] = U1[dof[t]];
    }
    if (lane < 10 * ncell) sJ[lane / 10][lane % 10] = ea.geom[c0 * 10 + lane];      // geom of consecutive cells is contiguous
    __syncthreads();
    if (q < NQ && half < ncell) {
      const int64_t c = c0 + half;
      const int kind = ea.cell_kind[c], region = ea.cell_region[c];
      const double* J = sJ[half];
      Kin<double> s, o;
      interpolate2(sU[half], sU1[half], J, sT[q], sL[q], s, o);
      Slots<double> out;
      if (kind == 0) fluid_flux<double, PART_BOTH>(ep.fluid[region], ep.sc, s, o, out);
      else solid_flux<double, PART_BOTH>(ep.solid[region], ep.sc, s, o, out);
      const double w = J[9] * sW[q];
      double* S = sS[half][q];
      for (int i = 0; i < 3; ++i) {
        S[i] = w * out.dval[i];
        S[12 + i] = w * out.vval[i];
        for (int k = 0; k < 3; ++k) {     // gradient slots pulled back to reference coordinates
          S[3 + 3 * i + k] = w * (out.dgrd[i][0] * J[3 * k] + out.dgrd[i][1] * J[3 * k + 1] + out.dgrd[i][2] * J[3 * k + 2]);
          S[15 + 3 * i + k] = w * (out.vgrd[i][0] * J[3 * k] + out.vgrd[i][1] * J[3 * k + 1] + out.vgrd[i][2] * J[3 * k + 2]);
        }
      }
      S[24] = w * out.pval;
    }
    __syncthreads();
    for (int t = 0; t < ncell; ++t) {
      double r = 0.0;
      if (lane < 60) {
        const int fld = lane / 30, comp = (lane % 30) / 10, a = lane % 10;
        const int vo = fld * 12 + comp, go = fld * 12 + 3 + 3 * comp;
        for (int k = 0; k < NQ; ++k) {
          const double4 tb = sT[k][a];
          r += sS[t][k][vo] * tb.x + sS[t][k][go] * tb.y + sS[t][k][go + 1] * tb.z + sS[t][k][go + 2] * tb.w;
        }
      } else {
        const int a = lane - 60;
        for (int k = 0; k < NQ; ++k) r += sS[t][k][24] * sL[k][a];
      }
      if (Re) Re[(c0 + t) * NLOC + (lane < 60 ? 6 * (lane % 10) + 3 * (lane / 30) + (lane % 30) / 10 : lane)] = r;
      else unsafeAtomicAdd(&F[dof[t]], r);
    }
  }
}
// F[dof] = sum over the cells around the dof's node of that cell's entry, in ascending cell order (fixed by the host).
// One thread per dof; the six threads of a node read six neighbouring doubles of every incident cell's vector.
// inc[k] = 16 * cell + local node of the k-th incidence of a node (ranks, inc_ptr) / of a pressure row (pinc_ptr, pinc).
__global__ __launch_bounds__(256) void k_residual_gather(int64_t N2, int64_t V, const int64_t* __restrict__ inc_ptr,
                                                         const int32_t* __restrict__ inc, const int64_t* __restrict__ pinc_ptr,
                                                         const int32_t* __restrict__ pinc, const double* __restrict__ Re,
                                                         double* __restrict__ F) {
  // A vertex has ~24 incident cells, an edge node ~5, and a wave holds both: with one dependent pair of loads (incidence ->
  // entry) per loop trip the wave would wait 24 memory latencies in a row.  Eight incidences per trip: eight independent
  // index loads, then eight independent entry loads, then the eight adds in ascending order (the order - and so the sum - is
  // the same whatever the grouping).
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= 6 * N2 + V) return;
  const bool node = i < 6 * N2;
  const int64_t r = node ? i / 6 : i - 6 * N2;
  const int off = node ? (int)(i - 6 * r) : 60;
  const int mul = node ? 6 : 1;
  const int64_t* ptr = node ? inc_ptr : pinc_ptr;
  const int32_t* lst = node ? inc : pinc;
  const int64_t k1 = ptr[r + 1];
  double s = 0.0;
  for (int64_t k = ptr[r]; k < k1; k += 8) {
    int32_t e[8];
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = k + j < k1 ? lst[k + j] : -1;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = e[j] >= 0 ? Re[(int64_t)(e[j] >> 4) * NLOC + mul * (e[j] & 15) + off] : 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  F[i] = s;
}
// ---------------------------------------------------------------------------------------------------------
// L2(Omega) norm of a mixed function: out += int |d|^2 + |v|^2 + p^2 dx over the cells  (`norm(dvp_res, 'l2')` of the
// reference's newtonsolver is DOLFIN's *function* norm; see oracle/fsi_oracle.py:function_norm)
// ---------------------------------------------------------------------------------------------------------
// One wave per cell, waves stride over the cells and keep their sum in a register; a workgroup's four sums go to
// part[blockIdx.x] and one further workgroup adds the partials in a fixed order (k_sum_parts): the same cells meet the same
// partial sums in every run, so the norm is bitwise reproducible (atomics on one address were neither that nor cheap: a
// million same-address atomics cost more than the whole integration).  Values only - no gradients are needed.
__global__ __launch_bounds__(256) void k_l2norm(ElemArrays ea, const double* __restrict__ X, int64_t C, double* __restrict__ part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  __shared__ double sU[4][NLOC];
  __shared__ double sAcc[4];
  double acc = 0.0;
  for (int64_t c = wave; c < C; c += nwaves) {
    sU[w][lane] = X[ea.cell_dofs[c * NLOC + lane]];
    __builtin_amdgcn_wave_barrier();
    if (lane < NQ) {
      const double* U = sU[w];
      double s = 0.0;
      for (int f = 0; f < 6; ++f) {                        // d_x d_y d_z v_x v_y v_z: 10 P2 values each
        double val = 0.0;
        for (int a = 0; a < 10; ++a) val += c_N[lane][a] * U[10 * f + a];
        s += val * val;
      }
      double p = 0.0;
      for (int a = 0; a < 4; ++a) p += c_L[lane][a] * U[60 + a];
      acc += ea.geom[c * 10 + 9] * c_qw[lane] * (s + p * p);
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) sAcc[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sAcc[0] + sAcc[1]) + (sAcc[2] + sAcc[3]);
}
// out[0] = sum of part[0 .. n) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void k_sum_parts(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// ---------------------------------------------------------------------------------------------------------
// per-step diagnostics of post_solve [REF src/vasp/simulations/simulation_common.py:253-348]: per cell the DG0
// projection (= quadrature mean) of |v| and of det(I + grad d); and point evaluation of (d, v, p) at located probes
// [REF :157-222].  cellvals: [2][C].
// ---------------------------------------------------------------------------------------------------------
// Four waves per workgroup, each striding over the cells (round 2: one 64-lane workgroup per cell, 1.29 ms on 1.12 M cells per
// time step): only what the two statistics need is interpolated - v and grad d -, the state is gathered through the node ranks.
__global__ __launch_bounds__(256) void k_cell_stats(ElemArrays ea, const double* __restrict__ X, int64_t C,
                                                    double* __restrict__ cellvals) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  __shared__ double sU[4][NLOC], sJ[4][10];
  for (int64_t c = wave; c < C; c += nwaves) {
    sU[w][lane] = X[lane < 60 ? 6 * ea.cell_rank[c * 10 + lane % 10] + 3 * (lane / 30) + (lane % 30) / 10 : ea.cell_prow[c * 4 + lane - 60]];
    if (lane < 10) sJ[w][lane] = ea.geom[c * 10 + lane];
    __builtin_amdgcn_wave_barrier();
    double sv = 0.0, sj = 0.0;
    if (lane < NQ) {
      const double* U = sU[w];
      const double* Jinv = sJ[w];
      double v[3] = {0.0, 0.0, 0.0}, gd[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
      for (int a = 0; a < 10; ++a) {
        const double N = c_N[lane][a];
        const double r0 = c_dN[lane][a][0], r1 = c_dN[lane][a][1], r2 = c_dN[lane][a][2];
        double G[3];
        for (int jj = 0; jj < 3; ++jj) G[jj] = r0 * Jinv[jj] + r1 * Jinv[3 + jj] + r2 * Jinv[6 + jj];
        for (int i = 0; i < 3; ++i) {
          v[i] += N * U[30 + i * 10 + a];
          const double dv = U[i * 10 + a];
          for (int jj = 0; jj < 3; ++jj) gd[i][jj] += dv * G[jj];
        }
      }
      const double wq = 6.0 * c_qw[lane];                      // weights sum to 1/6: cell mean
      sv = wq * sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      double Fi[3][3];
      sj = wq * inv_det_F<double>(gd, Fi);
    }
    for (int off = 32; off > 0; off >>= 1) { sv += __shfl_xor(sv, off, 64); sj += __shfl_xor(sj, off, 64); }
    if (lane == 0) { cellvals[c] = sv; cellvals[C + c] = sj; }
    __builtin_amdgcn_wave_barrier();
  }
}
// out[0..3] = sum, min, max of a[0..n) ; out[4] = min of b[0..n).  One block; n up to a few million.
// Two stages (deterministic): gridDim.x workgroups each reduce a strided part into part[4 * block ..], then one workgroup
// reduces the partials (stage 2: a = part, stride 4, n = number of partials; b unused).  Round 2: one workgroup for
// everything, 0.43 ms per time step.
__global__ __launch_bounds__(1024) void k_stats_reduce(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                       double* __restrict__ out, int stage) {
  __shared__ double ssum[1024], smin[1024], smax[1024], sminb[1024];
  double s = 0.0, mn = 1e300, mx = -1e300, mb = 1e300;
  if (stage == 2) {
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
      s += a[4 * i]; mn = fmin(mn, a[4 * i + 1]); mx = fmax(mx, a[4 * i + 2]); mb = fmin(mb, a[4 * i + 3]);
    }
  } else
  for (int64_t i = blockIdx.x * (int64_t)1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
    const double v = a[i];
    s += v; mn = fmin(mn, v); mx = fmax(mx, v); mb = fmin(mb, b[i]);
  }
  ssum[threadIdx.x] = s; smin[threadIdx.x] = mn; smax[threadIdx.x] = mx; sminb[threadIdx.x] = mb;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + off]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + off]);
      sminb[threadIdx.x] = fmin(sminb[threadIdx.x], sminb[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = stage == 1 ? out + 4 * blockIdx.x : out;
    o[0] = ssum[0]; o[1] = smin[0]; o[2] = smax[0]; o[3] = sminb[0];
  }
}
// probes: out[i][0..6] = d(3), v(3), p at barycentric coordinates bary[i][4] of cell cells[i] (P2 / P1 interpolation)
__global__ void k_probe(int64_t n, ElemArrays ea, const int32_t* __restrict__ cells, const double* __restrict__ bary,
                        const double* __restrict__ X, double* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cells[i];
  const double* l = bary + 4 * i;
  double N[10];
  for (int a = 0; a < 4; ++a) N[a] = l[a] * (2.0 * l[a] - 1.0);
  const int E[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  for (int e = 0; e < 6; ++e) N[4 + e] = 4.0 * l[E[e][0]] * l[E[e][1]];
  const int32_t* dofs = ea.cell_dofs + c * NLOC;
  for (int f = 0; f < 6; ++f) {
    double s = 0.0;
    for (int a = 0; a < 10; ++a) s += N[a] * X[dofs[f * 10 + a]];
    out[7 * i + f] = s;
  }
  double p = 0.0;
  for (int a = 0; a < 4; ++a) p += l[a] * X[dofs[60 + a]];
  out[7 * i + 6] = p;
}
void launch_cell_stats(hipStream_t st, int64_t C, const ElemArrays& ea, const double* X, double* cellvals, double* out) {
  const int64_t blocks = std::min<int64_t>((C + 3) / 4, 8192);
  hipLaunchKernelGGL(k_cell_stats, dim3((unsigned)blocks), dim3(256), 0, st, ea, X, C, cellvals);
  // `out` has room for 8 doubles; the partials of stage 1 live behind the two value arrays' consumer-visible part: the caller
  // passes cellvals with 2 C + 8 + 4 * STAT_PARTS doubles
  const int parts = (int)std::min<int64_t>(STAT_PARTS, std::max<int64_t>(1, (C + 1023) / 1024));
  double* part = out + 8;
  hipLaunchKernelGGL(k_stats_reduce, dim3(parts), dim3(1024), 0, st, C, cellvals, cellvals + C, part, 1);
  hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, st, (int64_t)parts, part, nullptr, out, 2);
}
void launch_probe(hipStream_t st, int64_t n, const ElemArrays& ea, const int32_t* cells, const double* bary, const double* X,
                  double* out) {
  hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, ea, cells, bary, X, out);
}
// part: room for 4096 partial sums; out[0] is overwritten (not accumulated)
void launch_l2norm(hipStream_t st, int64_t C, const ElemArrays& ea, const double* X, double* part, double* out) {
  const int64_t blocks = std::min<int64_t>((C + 3) / 4, 4096);
  hipLaunchKernelGGL(k_l2norm, dim3((unsigned)blocks), dim3(256), 0, st, ea, X, C, part);
  hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(256), 0, st, part, (int)blocks, out);
}

// ---------------------------------------------------------------------------------------------------------
// Jacobian: vals += d(element vector)/d(local dofs), PART selects F_linear (A_pre) or F_nonlinear
// ---------------------------------------------------------------------------------------------------------
// Rows that a part cannot touch are not carried: F_nonlinear has no d-equation terms (the Laplace lifting and the solid's
// displacement equation are linear: they live in A_pre), so the refresh kernel accumulates and scatters rows 30..63 only
// (34 instead of 64 accumulators per lane, half the contraction); on a solid cell F_nonlinear = theta0 (P(d), grad psi)
// depends on d alone, so only the 30 d-columns (lanes 0..29) do any work.  `list` as for k_residual: the cells of one colour.
template <int PART, int WAVES>     // WAVES: waves per SIMD the register budget is set for
__global__ __launch_bounds__(64, WAVES) void k_jacobian(ElemArrays ea, ElemParams ep, const double* __restrict__ U,
                                                 const double* __restrict__ U1, const int64_t* __restrict__ rowptr,
                                                 const int64_t* __restrict__ nadj_ptr, double* __restrict__ vals,
                                                 const int32_t* __restrict__ list) {
  constexpr int R0 = (PART == PART_NONLINEAR) ? 30 : 0;      // first local row this part can produce
  constexpr int NR = NLOC - R0;
  const int64_t c = list ? (int64_t)list[blockIdx.x] : (int64_t)blockIdx.x;
  const int lane = threadIdx.x;
  __shared__ double sU[NLOC], sU1[NLOC], sJ[10];
  __shared__ double sG[NQ][10][3];       // physical gradients of the P2 basis at the quadrature points
  __shared__ double sB[NQ][25];          // base state at n: gd(9) gv(9) d(3) v(3) p
  __shared__ double sO[NQ][25];          // state at n-1
  __shared__ int64_t sRow[NLOC];
  __shared__ int32_t sDeg6[10];
  const int32_t dof = ea.cell_dofs[c * NLOC + lane];
  sU[lane] = U[dof];
  sU1[lane] = U1[dof];
  sRow[lane] = rowptr[dof];
  if (lane < 10) {
    sJ[lane] = ea.geom[c * 10 + lane];
    const int32_t rk = ea.cell_rank[c * 10 + lane];
    sDeg6[lane] = 6 * (int32_t)(nadj_ptr[rk + 1] - nadj_ptr[rk]);
  }
  __syncthreads();
  for (int t = lane; t < NQ * 10; t += 64) {
    const int q = t / 10, a = t % 10;
    for (int j = 0; j < 3; ++j)
      sG[q][a][j] = c_dN[q][a][0] * sJ[j] + c_dN[q][a][1] * sJ[3 + j] + c_dN[q][a][2] * sJ[6 + j];
  }
  if (lane < NQ) {
    Kin<double> s, o;
    interpolate(sU, sJ, lane, s);
    interpolate(sU1, sJ, lane, o);
    double* B = sB[lane];
    double* O = sO[lane];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        B[3 * i + j] = s.gd[i][j]; B[9 + 3 * i + j] = s.gv[i][j];
        O[3 * i + j] = o.gd[i][j]; O[9 + 3 * i + j] = o.gv[i][j];
      }
    for (int i = 0; i < 3; ++i) { B[18 + i] = s.d[i]; B[21 + i] = s.v[i]; O[18 + i] = o.d[i]; O[21 + i] = o.v[i]; }
    B[24] = s.p; O[24] = o.p;
  }
  __syncthreads();
  const int kind = ea.cell_kind[c], region = ea.cell_region[c];
  // trial dof of this lane
  const int jf = lane < 60 ? lane / 30 : 2;          // 0 d, 1 v, 2 p
  const int jc = lane < 60 ? (lane % 30) / 10 : 0;
  const int jb = lane < 60 ? lane % 10 : lane - 60;
  if (PART == PART_NONLINEAR && kind == 1 && jf != 0) return;    // no barrier below
  double acc[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) acc[i] = 0.0;
  for (int q = 0; q < NQ; ++q) {
    Kin<Dual> s;
    Kin<double> o;
    const double* B = sB[q];
    const double* O = sO[q];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        s.gd[i][j] = Dual(B[3 * i + j], (jf == 0 && jc == i) ? sG[q][jb][j] : 0.0);
        s.gv[i][j] = Dual(B[9 + 3 * i + j], (jf == 1 && jc == i) ? sG[q][jb][j] : 0.0);
        o.gd[i][j] = O[3 * i + j];
        o.gv[i][j] = O[9 + 3 * i + j];
      }
    for (int i = 0; i < 3; ++i) {
      s.d[i] = Dual(B[18 + i], (jf == 0 && jc == i) ? c_N[q][jb] : 0.0);
      s.v[i] = Dual(B[21 + i], (jf == 1 && jc == i) ? c_N[q][jb] : 0.0);
      o.d[i] = O[18 + i];
      o.v[i] = O[21 + i];
    }
    s.p = Dual(B[24], jf == 2 ? c_L[q][jb] : 0.0);
    o.p = O[24];
    Slots<Dual> out;
    if (kind == 0) fluid_flux<Dual, PART>(ep.fluid[region], ep.sc, s, o, out);
    else solid_flux<Dual, PART>(ep.solid[region], ep.sc, s, o, out);
    const double w = sJ[9] * c_qw[q];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double vv = w * out.vval[i].e;
      const double v0 = w * out.vgrd[i][0].e, v1 = w * out.vgrd[i][1].e, v2 = w * out.vgrd[i][2].e;
      if (R0 == 0) {
        const double dv = w * out.dval[i].e;
        const double d0 = w * out.dgrd[i][0].e, d1 = w * out.dgrd[i][1].e, d2 = w * out.dgrd[i][2].e;
#pragma unroll
        for (int a = 0; a < 10; ++a)
          acc[i * 10 + a] += dv * c_N[q][a] + d0 * sG[q][a][0] + d1 * sG[q][a][1] + d2 * sG[q][a][2];
      }
#pragma unroll
      for (int a = 0; a < 10; ++a)
        acc[30 - R0 + i * 10 + a] += vv * c_N[q][a] + v0 * sG[q][a][0] + v1 * sG[q][a][1] + v2 * sG[q][a][2];
    }
    const double pv = w * out.pval.e;
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[60 - R0 + a] += pv * c_L[q][a];
  }
  // scatter column `lane` of the element matrix
  const uint16_t* nb = ea.enbr + c * 100;
  const uint16_t* pb = ea.epnbr + c * 40;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int i = R0 + k;
    const int ra = i < 60 ? i % 10 : i - 60;
    int64_t pos;
    if (jf < 2) pos = sRow[i] + 6 * (int64_t)nb[ra * 10 + jb] + 3 * jf + jc;
    else pos = sRow[i] + sDeg6[ra] + pb[ra * 4 + jb];
    if (acc[k] != 0.0) unsafeAtomicAdd(&vals[pos], acc[k]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Jacobian of F_nonlinear with the element contraction on the matrix pipe (BASELINE north_star: "MFMA only for the dense
// per-element contractions").
//
// The v-equation rows of the element matrix are, per component i,  D_i[a][j] = sum_q sum_s T_q[a][s] S_q,i[s][j]:
// test node a (10), trial column j (64), s = (value, 3 gradient slots), T_q[a] = (N_a, grad N_a) at point q and S the
// weighted tangent of the flux along column j - a [10 x 96] x [96 x 64] product per component, 75 % of the flops of the
// 64-row kernel of round 2 and 35 % of this one (rows 30..63 only).  On v_mfma_f64_16x16x4_f64 tiles: M = test nodes (10
// of 16 rows), N = 16 columns per tile (four tiles), K = the four slots of one point: 12 MFMAs per point, 288 per cell, the
// accumulators (12 tiles x 4 doubles per lane) live in the MFMA's own result registers.
//   A operand: lane l holds T_q[a = l & 15][s = l >> 4]: one LDS read per point from sA[q][s][a] (the table the flux phase
//              also takes its tangent seeds from).
//   B operand of tile t: lane l needs S[s = l >> 4][column 16 t + (l & 15)], while the flux phase leaves column l's four
//              slots in four registers of lane l: a 4 x 4 transpose between registers and the wave's four 16-lane rows,
//              done in registers with gfx950's v_permlane32_swap / v_permlane16_swap (4 swaps per 32-bit half; no LDS).
//   D: lane l holds rows (l >> 4) + 4 r, r = 0..3, of column 16 t + (l & 15): the scatter walks those.
// The four pressure rows (one slot, K = 24) stay on the vector pipe.  tools/mfma_layout_check.hip checks both register maps
// on the device; the parity tests compare the assembled matrix with the oracle's complex-step Jacobian entry by entry.
// ---------------------------------------------------------------------------------------------------------
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef double v4d_t __attribute__((ext_vector_type(4)));
__device__ inline void lane_swap32(double& x, double& y) {      // rows 2, 3 of x <-> rows 0, 1 of y
  const unsigned xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
  const v2u_t lo = __builtin_amdgcn_permlane32_swap(xl, yl, false, false), hi = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
  x = __hiloint2double(hi[0], lo[0]); y = __hiloint2double(hi[1], lo[1]);
}
__device__ inline void lane_swap16(double& x, double& y) {      // rows 1, 3 of x <-> rows 0, 2 of y
  const unsigned xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
  const v2u_t lo = __builtin_amdgcn_permlane16_swap(xl, yl, false, false), hi = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
  x = __hiloint2double(hi[0], lo[0]); y = __hiloint2double(hi[1], lo[1]);
}
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_jacobian_mfma(ElemArrays ea, ElemParams ep, const double* __restrict__ U,
                                                             const double* __restrict__ U1, const int64_t* __restrict__ rowptr,
                                                             const int64_t* __restrict__ nadj_ptr, double* __restrict__ vals,
                                                             const int32_t* __restrict__ list) {
  const int64_t c = list ? (int64_t)list[blockIdx.x] : (int64_t)blockIdx.x;
  const int lane = threadIdx.x;
  __shared__ double sU[NLOC], sU1[NLOC], sJ[10];
  __shared__ double sA[NQ][4][10];       // (N, physical gradient) of the P2 basis at the quadrature points: [q][slot][node]
  __shared__ double sB[NQ][25];          // base state at n: gd(9) gv(9) d(3) v(3) p
  __shared__ double sO[NQ][25];          // state at n-1
  __shared__ int64_t sRow[NLOC];
  __shared__ int32_t sDeg6[10];
  const int32_t dof = ea.cell_dofs[c * NLOC + lane];
  sU[lane] = U[dof];
  sU1[lane] = U1[dof];
  sRow[lane] = rowptr[dof];
  if (lane < 10) {
    sJ[lane] = ea.geom[c * 10 + lane];
    const int32_t rk = ea.cell_rank[c * 10 + lane];
    sDeg6[lane] = 6 * (int32_t)(nadj_ptr[rk + 1] - nadj_ptr[rk]);
  }
  __syncthreads();
  for (int t = lane; t < NQ * 10; t += 64) {
    const int q = t / 10, a = t % 10;
    sA[q][0][a] = c_N[q][a];
    for (int j = 0; j < 3; ++j)
      sA[q][1 + j][a] = c_dN[q][a][0] * sJ[j] + c_dN[q][a][1] * sJ[3 + j] + c_dN[q][a][2] * sJ[6 + j];
  }
  if (lane < NQ) {
    Kin<double> s, o;
    interpolate(sU, sJ, lane, s);
    interpolate(sU1, sJ, lane, o);
    double* B = sB[lane];
    double* O = sO[lane];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        B[3 * i + j] = s.gd[i][j]; B[9 + 3 * i + j] = s.gv[i][j];
        O[3 * i + j] = o.gd[i][j]; O[9 + 3 * i + j] = o.gv[i][j];
      }
    for (int i = 0; i < 3; ++i) { B[18 + i] = s.d[i]; B[21 + i] = s.v[i]; O[18 + i] = o.d[i]; O[21 + i] = o.v[i]; }
    B[24] = s.p; O[24] = o.p;
  }
  __syncthreads();
  const int kind = ea.cell_kind[c], region = ea.cell_region[c];
  const int jf = lane < 60 ? lane / 30 : 2;          // trial dof of this lane: 0 d, 1 v, 2 p
  const int jc = lane < 60 ? (lane % 30) / 10 : 0;
  const int jb = lane < 60 ? lane % 10 : lane - 60;
  // a solid cell's F_nonlinear depends on d alone: the other columns carry zeros through the product (the wave stays whole
  // for the MFMAs) and are skipped by the scatter
  const bool live = !(kind == 1 && jf != 0);
  const int am = lane & 15, ak = lane >> 4;          // this lane's element of the A operand: node am, slot ak
  v4d_t acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = v4d_t{0.0, 0.0, 0.0, 0.0};
  double accp[4] = {0.0, 0.0, 0.0, 0.0};
  for (int q = 0; q < NQ; ++q) {
    Kin<Dual> s;
    Kin<double> o;
    const double* B = sB[q];
    const double* O = sO[q];
    const double seedN = sA[q][0][jb];
    double seedG[3];
    for (int j = 0; j < 3; ++j) seedG[j] = sA[q][1 + j][jb];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        s.gd[i][j] = Dual(B[3 * i + j], (jf == 0 && jc == i) ? seedG[j] : 0.0);
        s.gv[i][j] = Dual(B[9 + 3 * i + j], (jf == 1 && jc == i) ? seedG[j] : 0.0);
        o.gd[i][j] = O[3 * i + j];
        o.gv[i][j] = O[9 + 3 * i + j];
      }
    for (int i = 0; i < 3; ++i) {
      s.d[i] = Dual(B[18 + i], (jf == 0 && jc == i) ? seedN : 0.0);
      s.v[i] = Dual(B[21 + i], (jf == 1 && jc == i) ? seedN : 0.0);
      o.d[i] = O[18 + i];
      o.v[i] = O[21 + i];
    }
    s.p = Dual(B[24], jf == 2 ? c_L[q][jb] : 0.0);
    o.p = O[24];
    Slots<Dual> out;
    if (kind == 0) fluid_flux<Dual, PART_NONLINEAR>(ep.fluid[region], ep.sc, s, o, out);
    else solid_flux<Dual, PART_NONLINEAR>(ep.solid[region], ep.sc, s, o, out);
    const double w = live ? sJ[9] * c_qw[q] : 0.0;
    const double aop = am < 10 ? sA[q][ak][am] : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double b0 = w * out.vval[i].e, b1 = w * out.vgrd[i][0].e, b2 = w * out.vgrd[i][1].e, b3 = w * out.vgrd[i][2].e;
      // registers <-> 16-lane rows: afterwards b_t of lane (row g, n) = slot g of column 16 t + n
      lane_swap32(b0, b2); lane_swap32(b1, b3); lane_swap16(b0, b1); lane_swap16(b2, b3);
      acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, b0, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, b1, acc[i][1], 0, 0, 0);
      acc[i][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, b2, acc[i][2], 0, 0, 0);
      acc[i][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, b3, acc[i][3], 0, 0, 0);
    }
    const double pv = w * out.pval.e;
#pragma unroll
    for (int a = 0; a < 4; ++a) accp[a] += pv * c_L[q][a];
  }
  const uint16_t* nb = ea.enbr + c * 100;
  const uint16_t* pb = ea.epnbr + c * 40;
  // v rows: tile t holds columns 16 t + (lane & 15); this lane's four results are test nodes (lane >> 4) + 4 r
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = 16 * t + am;
    const int cf = col < 60 ? col / 30 : 2, cc = col < 60 ? (col % 30) / 10 : 0, cb = col < 60 ? col % 10 : col - 60;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ra = ak + 4 * r;
      if (ra >= 10) continue;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double v = acc[i][t][r];
        if (v == 0.0) continue;
        const int row = 30 + 10 * i + ra;
        const int64_t pos = cf < 2 ? sRow[row] + 6 * (int64_t)nb[ra * 10 + cb] + 3 * cf + cc : sRow[row] + sDeg6[ra] + pb[ra * 4 + cb];
        unsafeAtomicAdd(&vals[pos], v);
      }
    }
  }
  // p rows: column `lane`
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (accp[a] == 0.0) continue;
    const int row = 60 + a;
    const int64_t pos = jf < 2 ? sRow[row] + 6 * (int64_t)nb[a * 10 + jb] + 3 * jf + jc : sRow[row] + sDeg6[a] + pb[a * 4 + jb];
    unsafeAtomicAdd(&vals[pos], accp[a]);
  }
}

void launch_geometry(hipStream_t st, int64_t C, const double* coords, const int32_t* tet_vertices, double* geom) {
  const int bs = 256;
  hipLaunchKernelGGL(k_geometry, dim3((unsigned)((C + bs - 1) / bs)), dim3(bs), 0, st, C, coords, tet_vertices, geom);
}
// gather form (Re and the incidence lists given): every entry of F is written, no memset needed; otherwise F += with atomics
void launch_residual(hipStream_t st, int64_t C, const ElemArrays& ea, const ElemParams& ep, const double* U,
                     const double* U1, double* F, const ResidualGather& rg) {
  const int64_t grid = std::min<int64_t>(xcd_grid((C + 1) / 2), 256 * 8 * 4);      // persistent: a few rounds of the resident workgroups
  hipLaunchKernelGGL(k_residual<2>, dim3((unsigned)grid), dim3(64), 0, st, ea, ep, U, U1, F, C, rg.Re);
  if (rg.Re) {
    const int64_t n = 6 * rg.N2 + rg.V;
    hipLaunchKernelGGL(k_residual_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rg.N2, rg.V, rg.inc_ptr, rg.inc,
                       rg.pinc_ptr, rg.pinc, rg.Re, F);
  }
}
// `cc`: the assembly colouring (cells sorted by colour; colour k = cells[ptr[k] .. ptr[k+1])), or ncolours == 0 for one launch
// over all cells with atomics whose order varies from run to run.
void launch_jacobian(hipStream_t st, int part, int64_t C, const ElemArrays& ea, const ElemParams& ep, const double* U,
                     const double* U1, const int64_t* rowptr, const int64_t* nadj_ptr, double* vals, const CellColours& cc,
                     int jac_waves, int jac_mfma) {
  // jac_waves (FsiTuning.jacobian_waves): two waves per SIMD with 94 spilled VGPRs beat one wave without (70 against 99 ms per
  // refresh at 1.12 M tets); jac_mfma (FsiTuning.jacobian_mfma): k_jacobian_mfma for the refresh kernel (103 ms: not the default)
  const int rounds = cc.ncolours == 0 ? 1 : cc.ncolours;
  for (int k = 0; k < rounds; ++k) {
    const int64_t n = cc.ncolours == 0 ? C : cc.ptr[k + 1] - cc.ptr[k];
    const int32_t* list = cc.ncolours == 0 ? nullptr : cc.cells + cc.ptr[k];
    if (n <= 0) continue;
    if (part == PART_LINEAR)
      hipLaunchKernelGGL((k_jacobian<PART_LINEAR, 1>), dim3((unsigned)n), dim3(64), 0, st, ea, ep, U, U1, rowptr, nadj_ptr, vals, list);
    else if (jac_mfma && jac_waves == 2)
      hipLaunchKernelGGL((k_jacobian_mfma<2>), dim3((unsigned)n), dim3(64), 0, st, ea, ep, U, U1, rowptr, nadj_ptr, vals, list);
    else if (jac_mfma)
      hipLaunchKernelGGL((k_jacobian_mfma<1>), dim3((unsigned)n), dim3(64), 0, st, ea, ep, U, U1, rowptr, nadj_ptr, vals, list);
    else if (jac_waves == 2)
      hipLaunchKernelGGL((k_jacobian<PART_NONLINEAR, 2>), dim3((unsigned)n), dim3(64), 0, st, ea, ep, U, U1, rowptr, nadj_ptr, vals, list);
    else
      hipLaunchKernelGGL((k_jacobian<PART_NONLINEAR, 1>), dim3((unsigned)n), dim3(64), 0, st, ea, ep, U, U1, rowptr, nadj_ptr, vals, list);
  }
}

}  // namespace fsi
