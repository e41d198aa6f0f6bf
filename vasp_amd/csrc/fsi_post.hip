// Solid stress / strain and wall shear stress from the resident state.
//
// Replaces, per saved time step, the element loops of VaSP's FEniCS post-processing:
//   k_stress_strain : compute_stress_strain [REF src/vasp/postprocessing/postprocessing_fenics/compute_stress_strain.py:188-263]
//                     Cauchy stress 1/J F S F^T and Green-Lagrange strain E of the P2 displacement, L2-projected onto
//                     tensor DG1 cell by cell (solve_dg), then the largest principal value of each projected tensor
//                     (common.get_eig) projected onto scalar DG1 (project_dg).  The constitutive routines are those of
//                     the residual kernels (fsi_element.hpp).
//   k_wss           : Stress of compute_hemodynamics [REF .../compute_hemodynamics.py:91-157]: Ft = F - (F.n) n with
//                     F = -2 mu sym(grad u) n on exterior facets, projected with the surface mass matrix onto the DG1
//                     space of the boundary cell (zero rows -> identity).
// Both are cell-local: one wavefront per solid cell (lanes = quadrature points, LDS for the projections), one lane per
// boundary cell.  Output is DG1 coefficients (one per local vertex), the layout the reference's write_checkpoint files
// carry through cell_dofs.  HBM traffic per solid cell: 30 gathered doubles + 80 geometry bytes in, 80 doubles out.
#include "fsi_kernels.hpp"

namespace fsi {

namespace {

// Keast-24 tables as in fsi_assembly.hip (this translation unit keeps its own constant copies)
__constant__ double p_qw[NQ];
__constant__ double p_dN[NQ][10][3];
__constant__ double p_L[NQ][4];
bool p_tables_ready = false;

__device__ inline double max_eig_sym3(const double T[3][3]) {
  // largest root of the characteristic polynomial, trigonometric form (Kopp 2008, eqs. 21-34) with the perturbations of
  // turtleFSI's get_eig so that p, q and the discriminant never vanish
  const double I1 = T[0][0] + T[1][1] + T[2][2];
  double TT = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) TT += T[i][j] * T[i][j];
  const double I2 = 0.5 * (I1 * I1 - TT);
  const double I3 = T[0][0] * (T[1][1] * T[2][2] - T[1][2] * T[2][1]) - T[0][1] * (T[1][0] * T[2][2] - T[1][2] * T[2][0]) +
                    T[0][2] * (T[1][0] * T[2][1] - T[1][1] * T[2][0]);
  double p = I1 * I1 - 3.0 * I2;
  if (p < 1e-16) p = fabs(p) + 2e-16;
  double q = 13.5 * I3 + I1 * I1 * I1 - 4.5 * I1 * I2;
  if (fabs(q) < 1e-24) q = q + (q > 0.0 ? 2e-24 : (q < 0.0 ? -2e-24 : 0.0));
  double nom2 = 27.0 * (0.25 * I2 * I2 * (p - I2) + I3 * (6.75 * I3 - q));
  if (nom2 < 1e-40) nom2 = fabs(nom2) + 2e-40;
  const double phi = atan2(sqrt(nom2), q) / 3.0;
  return (sqrt(p) * 2.0 * cos(phi) + I1) / 3.0;
}

// out[c][80]: TrueStress [4][9], GreenLagrangeStrain [4][9], MaxPrincipalStress [4], MaxPrincipalStrain [4]
__global__ __launch_bounds__(64) void k_stress_strain(ElemArrays ea, ElemParams ep, const double* __restrict__ U, int64_t ncell,
                                                      const int32_t* __restrict__ cells, double* __restrict__ out) {
  const int64_t ci = blockIdx.x;
  const int64_t c = cells[ci];
  const int lane = threadIdx.x;
  __shared__ double sD[30], sJ[10];
  __shared__ double sF[NQ][18];          // sigma(9), E(9) at the quadrature points, weighted
  __shared__ double sX[72];              // DG1 coefficients of the two tensors
  __shared__ double sP[NQ][2];           // principal values at the quadrature points, weighted
  if (lane < 30) sD[lane] = U[ea.cell_dofs[c * NLOC + lane]];
  if (lane < 10) sJ[lane] = ea.geom[c * 10 + lane];
  __syncthreads();
  const SolidProps sp = ep.solid[ea.cell_region[c]];
  if (lane < NQ) {
    double g[3][3] = {};
    for (int a = 0; a < 10; ++a) {
      const double r0 = p_dN[lane][a][0], r1 = p_dN[lane][a][1], r2 = p_dN[lane][a][2];
      double G[3];
      for (int j = 0; j < 3; ++j) G[j] = r0 * sJ[j] + r1 * sJ[3 + j] + r2 * sJ[6 + j];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g[i][j] += sD[i * 10 + a] * G[j];
    }
    double P[3][3], Fi[3][3];
    piola<double>(sp, g, P);                              // P = F S
    const double J = inv_det_F<double>(g, Fi);
    const double w = sJ[9] * p_qw[lane];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0, cij = 0.0;
        for (int k = 0; k < 3; ++k) {
          const double Fjk = g[j][k] + (j == k ? 1.0 : 0.0);
          s += P[i][k] * Fjk;                             // (F S) F^T
          cij += (g[k][i] + (k == i ? 1.0 : 0.0)) * (g[k][j] + (k == j ? 1.0 : 0.0));
        }
        sF[lane][3 * i + j] = w * s / J;
        sF[lane][9 + 3 * i + j] = w * 0.5 * (cij - (i == j ? 1.0 : 0.0));
      }
  }
  __syncthreads();
  // rhs_a = sum_q L[q][a] f_q ; the P1 mass matrix of a tetrahedron is vol/20 (I + 1 1^T), its inverse 20/vol (I - 1 1^T / 5)
  const double vol = sJ[9] / 6.0;
  for (int o = lane; o < 72; o += 64) {
    const int a = o / 18, comp = o % 18;
    double s = 0.0;
    for (int q = 0; q < NQ; ++q) s += p_L[q][a] * sF[q][comp];
    sX[o] = s;
  }
  __syncthreads();
  double keep[2] = {0.0, 0.0};
  for (int o = lane, k = 0; o < 72; o += 64, ++k) {
    const int comp = o % 18;
    const double tot = sX[comp] + sX[18 + comp] + sX[36 + comp] + sX[54 + comp];
    keep[k] = (20.0 / vol) * (sX[o] - 0.2 * tot);
  }
  __syncthreads();
  for (int o = lane, k = 0; o < 72; o += 64, ++k) sX[o] = keep[k];
  __syncthreads();
  double* oc = out + ci * 80;
  for (int o = lane; o < 72; o += 64) {
    const int a = o / 18, comp = o % 18;
    oc[(comp < 9 ? 0 : 36) + a * 9 + (comp % 9)] = sX[o];
  }
  if (lane < NQ) {
    const double w = sJ[9] * p_qw[lane];
    for (int t = 0; t < 2; ++t) {
      double T[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double s = 0.0;
          for (int a = 0; a < 4; ++a) s += p_L[lane][a] * sX[a * 18 + 9 * t + 3 * i + j];
          T[i][j] = s;
        }
      for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) T[i][j] = T[j][i] = 0.5 * (T[i][j] + T[j][i]);
      sP[lane][t] = w * max_eig_sym3(T);
    }
  }
  __syncthreads();
  if (lane < 8) {
    const int t = lane / 4, a = lane % 4;
    double r[4];
    for (int b = 0; b < 4; ++b) {
      double s = 0.0;
      for (int q = 0; q < NQ; ++q) s += p_L[q][b] * sP[q][t];
      r[b] = s;
    }
    oc[72 + 4 * t + a] = (20.0 / vol) * (r[a] - 0.2 * (r[0] + r[1] + r[2] + r[3]));
  }
}

// One lane per boundary cell: fmask bit f set = the facet opposite local vertex f is an exterior facet.
// out[c][4][3]: DG1 coefficients of the projected tangential traction (0 on vertices that touch no exterior facet).
__global__ __launch_bounds__(64) void k_wss(ElemArrays ea, const double* __restrict__ U, int64_t ncell,
                                            const int32_t* __restrict__ cells, const int32_t* __restrict__ fmask, double mu,
                                            double* __restrict__ out) {
  const int64_t ci = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (ci >= ncell) return;
  const int64_t c = cells[ci];
  const double* Jg = ea.geom + c * 10;
  double gl[4][3];                                  // physical gradients of the barycentric coordinates
  for (int j = 0; j < 3; ++j) {
    gl[1][j] = Jg[j]; gl[2][j] = Jg[3 + j]; gl[3][j] = Jg[6 + j];
    gl[0][j] = -(Jg[j] + Jg[3 + j] + Jg[6 + j]);
  }
  const double vol = Jg[9] / 6.0;
  double v[10][3];
  for (int i = 0; i < 3; ++i)
    for (int a = 0; a < 10; ++a) v[a][i] = U[ea.cell_dofs[c * NLOC + 30 + i * 10 + a]];
  const int E[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  // grad v is linear on the cell: its values at the four vertices
  double gv[4][3][3];
  for (int vtx = 0; vtx < 4; ++vtx)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int a = 0; a < 4; ++a) s += v[a][i] * ((a == vtx ? 3.0 : -1.0) * gl[a][j]);
        for (int e = 0; e < 6; ++e) {
          const int p = E[e][0], q = E[e][1];
          s += v[4 + e][i] * 4.0 * ((p == vtx ? 1.0 : 0.0) * gl[q][j] + (q == vtx ? 1.0 : 0.0) * gl[p][j]);
        }
        gv[vtx][i][j] = s;
      }
  double M[4][4] = {}, b[4][3] = {};
  const int mask = fmask[ci];
  for (int f = 0; f < 4; ++f) {
    if (!(mask & (1 << f))) continue;
    const double gn = sqrt(gl[f][0] * gl[f][0] + gl[f][1] * gl[f][1] + gl[f][2] * gl[f][2]);
    const double n[3] = {-gl[f][0] / gn, -gl[f][1] / gn, -gl[f][2] / gn};       // outward: away from the opposite vertex
    const double area = 3.0 * vol * gn;
    double Ft[4][3];
    for (int vtx = 0; vtx < 4; ++vtx) {
      if (vtx == f) continue;
      double Fv[3], Fn = 0.0;
      for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) s += mu * (gv[vtx][i][j] + gv[vtx][j][i]) * n[j];
        Fv[i] = -s;
        Fn += Fv[i] * n[i];
      }
      for (int i = 0; i < 3; ++i) Ft[vtx][i] = Fv[i] - Fn * n[i];
    }
    for (int a = 0; a < 4; ++a) {
      if (a == f) continue;
      for (int bb = 0; bb < 4; ++bb) {
        if (bb == f) continue;
        const double m = area / 12.0 * (a == bb ? 2.0 : 1.0);      // facet mass matrix of the P1 traces
        M[a][bb] += m;
        for (int i = 0; i < 3; ++i) b[a][i] += m * Ft[bb][i];
      }
    }
  }
  for (int a = 0; a < 4; ++a) {
    double s = 0.0;
    for (int bb = 0; bb < 4; ++bb) s += fabs(M[a][bb]);
    if (s == 0.0) M[a][a] = 1.0;                                    // ident_zeros of the surface mass matrix
  }
  // Gaussian elimination (symmetric positive definite after the identity rows)
  for (int k = 0; k < 4; ++k) {
    const double piv = 1.0 / M[k][k];
    for (int r = k + 1; r < 4; ++r) {
      const double l = M[r][k] * piv;
      if (l == 0.0) continue;
      for (int cc = k; cc < 4; ++cc) M[r][cc] -= l * M[k][cc];
      for (int i = 0; i < 3; ++i) b[r][i] -= l * b[k][i];
    }
  }
  for (int k = 3; k >= 0; --k)
    for (int i = 0; i < 3; ++i) {
      double s = b[k][i];
      for (int cc = k + 1; cc < 4; ++cc) s -= M[k][cc] * b[cc][i];
      b[k][i] = s / M[k][k];
    }
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 3; ++i) out[(ci * 4 + a) * 3 + i] = b[a][i];
}

}  // namespace

hipError_t upload_post_tables(const double* qw, const double* dN, const double* L) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(p_qw), qw, sizeof(double) * NQ);
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(p_dN), dN, sizeof(double) * NQ * 30);
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(p_L), L, sizeof(double) * NQ * 4);
  p_tables_ready = e == hipSuccess;
  return e;
}
void launch_stress_strain(hipStream_t st, int64_t ncell, const ElemArrays& ea, const ElemParams& ep, const double* U,
                          const int32_t* cells, double* out) {
  if (ncell > 0) hipLaunchKernelGGL(k_stress_strain, dim3((unsigned)ncell), dim3(64), 0, st, ea, ep, U, ncell, cells, out);
}
void launch_wss(hipStream_t st, int64_t ncell, const ElemArrays& ea, const double* U, const int32_t* cells, const int32_t* fmask,
                double mu, double* out) {
  if (ncell > 0)
    hipLaunchKernelGGL(k_wss, dim3((unsigned)((ncell + 63) / 64)), dim3(64), 0, st, ea, U, ncell, cells, fmask, mu, out);
}

}  // namespace fsi
