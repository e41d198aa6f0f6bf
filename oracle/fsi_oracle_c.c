/* CPU ORACLE (test infrastructure, not product code): plain-C restatement of the element routines, assembly and sparse
 * product of the monolithic ALE-FSI Newton step, OpenMP over the cells / rows.
 *
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this library; the product path
 * (vasp_amd, libvaspfsi.so) never does.  What is restated: see oracle/fsi_oracle.py (turtleFSI fluid.py / solid.py /
 * laplace.py / common.py for VaSP's call sites, DOLFIN assembly semantics); the numpy module is the definition, this file
 * is the same arithmetic in C so that (i) the CPU test-suite does not spend minutes in numpy complex-step loops and
 * (ii) bench.py has a CPU port of the algorithm that uses all host cores.  tests/test_oracle_c.py holds the two against
 * each other to round-off.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared; no dependencies)
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  const double* N;      /* [24][10]   P2 values at the Keast points                    */
  const double* dNref;  /* [24][10][3] reference gradients                             */
  const double* L;      /* [24][4]    P1 values                                        */
  const double* qw;     /* [24]       weights (sum 1/6)                                */
} OracleTables;

static void d_inv3(const double A[3][3], double Ai[3][3], double* det) {
  double c[3][3];
  c[0][0] = A[1][1] * A[2][2] - A[1][2] * A[2][1];
  c[0][1] = A[0][2] * A[2][1] - A[0][1] * A[2][2];
  c[0][2] = A[0][1] * A[1][2] - A[0][2] * A[1][1];
  c[1][0] = A[1][2] * A[2][0] - A[1][0] * A[2][2];
  c[1][1] = A[0][0] * A[2][2] - A[0][2] * A[2][0];
  c[1][2] = A[0][2] * A[1][0] - A[0][0] * A[1][2];
  c[2][0] = A[1][0] * A[2][1] - A[1][1] * A[2][0];
  c[2][1] = A[0][1] * A[2][0] - A[0][0] * A[2][1];
  c[2][2] = A[0][0] * A[1][1] - A[0][1] * A[1][0];
  const double d = A[0][0] * c[0][0] + A[0][1] * c[1][0] + A[0][2] * c[2][0];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ai[i][j] = c[i][j] / d;
  *det = d;
}

#define SC double
#define FN(x) r_##x
#define SC_POW(a, b) pow(a, b)
#define SC_LOG(a) log(a)
#include "fsi_oracle_elem.inc"
#undef SC
#undef FN
#undef SC_POW
#undef SC_LOG

#define SC double complex
#define FN(x) c_##x
#define SC_POW(a, b) cpow(a, b)
#define SC_LOG(a) clog(a)
#include "fsi_oracle_elem.inc"
#undef SC
#undef FN
#undef SC_POW
#undef SC_LOG

static const double* cell_props(int kind, int region, const double* fluid_props, const double* solid_props) {
  return kind == 0 ? fluid_props + 2 * region : solid_props + 6 * region;
}

/* Rl, Rn [C][64]: element residuals.  xc [C][4][3]; loc, loc1 [C][64] gathered states; kind/region [C];
 * fluid_props [nf][2] = (rho_f, mu_f); solid_props [ns][6] = (rho_s, mu_s, lambda_s, C10, C01, C11); solid_models [ns]. */
void fsi_c_element_residuals(int64_t C, const double* xc, const int32_t* kind, const int32_t* region,
                             const double* fluid_props, const double* solid_props, const int32_t* solid_models, double dt,
                             double theta, double delta, const double* N, const double* dNref, const double* L,
                             const double* qw, const double* loc, const double* loc1, double* Rl, double* Rn) {
  const OracleTables T = {N, dNref, L, qw};
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < C; ++c) {
    const int model = kind[c] == 1 ? solid_models[region[c]] : 0;
    r_element(xc + 12 * c, kind[c], cell_props(kind[c], region[c], fluid_props, solid_props), model, dt, theta, delta, &T,
              loc + 64 * c, loc1 + 64 * c, Rl + 64 * c, Rn + 64 * c);
  }
}

/* Jl, Jn [C][64][64]: d(Rl_e)/d(U^n_e), d(Rn_e)/d(U^n_e) by complex-step differentiation (h = 1e-30). */
void fsi_c_element_jacobians(int64_t C, const double* xc, const int32_t* kind, const int32_t* region,
                             const double* fluid_props, const double* solid_props, const int32_t* solid_models, double dt,
                             double theta, double delta, const double* N, const double* dNref, const double* L,
                             const double* qw, const double* loc, const double* loc1, double* Jl, double* Jn) {
  const OracleTables T = {N, dNref, L, qw};
  const double h = 1e-30;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t c = 0; c < C; ++c) {
    const int model = kind[c] == 1 ? solid_models[region[c]] : 0;
    const double* props = cell_props(kind[c], region[c], fluid_props, solid_props);
    double complex z[64], rl[64], rn[64];
    for (int a = 0; a < 64; ++a) z[a] = loc[64 * c + a];
    for (int j = 0; j < 64; ++j) {
      z[j] = loc[64 * c + j] + h * I;
      c_element(xc + 12 * c, kind[c], props, model, dt, theta, delta, &T, z, loc1 + 64 * c, rl, rn);
      z[j] = loc[64 * c + j];
      for (int a = 0; a < 64; ++a) {
        Jl[(64 * c + a) * 64 + j] = cimag(rl[a]) / h;
        Jn[(64 * c + a) * 64 + j] = cimag(rn[a]) / h;
      }
    }
  }
}

/* data[pos(row, col)] += Je[c][a][b] for row = cell_dofs[c][a], col = cell_dofs[c][b]; CSR with sorted column indices.
 * Returns the number of entries that were not found in the pattern (0 when the pattern is complete). */
int64_t fsi_c_scatter_csr(int64_t C, const int64_t* cell_dofs, const double* Je, const int64_t* indptr,
                          const int32_t* indices, double* data) {
  int64_t missing = 0;
#pragma omp parallel for schedule(static) reduction(+ : missing)
  for (int64_t c = 0; c < C; ++c) {
    const int64_t* dofs = cell_dofs + 64 * c;
    for (int a = 0; a < 64; ++a) {
      const int64_t row = dofs[a], s = indptr[row], e = indptr[row + 1];
      for (int b = 0; b < 64; ++b) {
        const double v = Je[(64 * c + a) * 64 + b];
        if (v == 0.0) continue;
        const int32_t col = (int32_t)dofs[b];
        int64_t lo = s, hi = e - 1;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (indices[mid] < col) lo = mid + 1; else hi = mid;
        }
        if (lo < e && indices[lo] == col) {
#pragma omp atomic
          data[lo] += v;
        } else {
          missing += 1;
        }
      }
    }
  }
  return missing;
}

/* F[dof] += Re[c][a] */
void fsi_c_scatter_vector(int64_t C, const int64_t* cell_dofs, const double* Re, double* F) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < C; ++c)
    for (int a = 0; a < 64; ++a) {
#pragma omp atomic
      F[cell_dofs[64 * c + a]] += Re[64 * c + a];
    }
}

/* y = A x (CSR) */
void fsi_c_spmv(int64_t n, const int64_t* indptr, const int32_t* indices, const double* data, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int64_t t = indptr[r]; t < indptr[r + 1]; ++t) s += data[t] * x[indices[t]];
    y[r] = s;
  }
}

int fsi_c_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
