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

/* ---- bench.py's CPU leg on the full bench mesh (oracle/cpu_port.py: full_mesh_kernels) ---------------------------------
 * The monolithic matrix's sparsity pattern straight from the P2 node graph, without scipy: DOLFIN's dofmap-based pattern
 * couples every field over the node graph (rows d / v of node a: the six d and v dofs of every neighbour node b and the
 * pressure dof of every neighbour VERTEX; pressure row of vertex a: the same columns).  Layout of FsiOracle: [d: 3 N2 | v:
 * 3 N2 | p: V], component-minor; vertices are the first V nodes. */

/* node graph: neighbours of node a = all nodes of the cells around a (a itself included), ascending.  inc_ptr / inc: cells
 * around each node.  Pass 1 (g_idx == NULL): deg[a].  Pass 2: g_idx filled at g_ptr[a]. */
void fsi_c_node_graph(int64_t N2, const int32_t* tn, const int64_t* inc_ptr, const int32_t* inc, const int64_t* g_ptr,
                      int32_t* g_idx, int32_t* deg) {
#pragma omp parallel
  {
    int32_t* buf = (int32_t*)malloc(4096 * sizeof(int32_t));
    int64_t cap = 4096;
#pragma omp for schedule(dynamic, 1024)
    for (int64_t a = 0; a < N2; ++a) {
      const int64_t nc = inc_ptr[a + 1] - inc_ptr[a];
      if (10 * nc > cap) { cap = 20 * nc; buf = (int32_t*)realloc(buf, cap * sizeof(int32_t)); }
      int64_t m = 0;
      for (int64_t t = inc_ptr[a]; t < inc_ptr[a + 1]; ++t)
        for (int k = 0; k < 10; ++k) buf[m++] = tn[10 * (int64_t)inc[t] + k];
      for (int64_t i = 1; i < m; ++i) {          /* insertion sort: ~250 entries, mostly runs */
        const int32_t v = buf[i];
        int64_t j = i - 1;
        while (j >= 0 && buf[j] > v) { buf[j + 1] = buf[j]; --j; }
        buf[j + 1] = v;
      }
      int64_t u = 0;
      for (int64_t i = 0; i < m; ++i)
        if (i == 0 || buf[i] != buf[i - 1]) buf[u++] = buf[i];
      if (g_idx) memcpy(g_idx + g_ptr[a], buf, u * sizeof(int32_t));
      else deg[a] = (int32_t)u;
    }
    free(buf);
  }
}

/* column indices (and deterministic stand-in values, first touched by the thread that will stream them) of the monolithic
 * CSR matrix on that pattern; indptr from the caller (row lengths 6 deg(a) + degV(a)). */
void fsi_c_monolithic_pattern(int64_t N2, int64_t V, const int64_t* g_ptr, const int32_t* g_idx, const int64_t* indptr,
                              int32_t* indices, double* data) {
  const int64_t n3 = 3 * N2, nrows = 6 * N2 + V;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    const int64_t a = r < 2 * n3 ? (r % n3) / 3 : r - 2 * n3;
    int64_t t = indptr[r];
    for (int blk = 0; blk < 2; ++blk)
      for (int64_t s = g_ptr[a]; s < g_ptr[a + 1]; ++s)
        for (int c = 0; c < 3; ++c) indices[t++] = (int32_t)(blk * n3 + 3 * (int64_t)g_idx[s] + c);
    for (int64_t s = g_ptr[a]; s < g_ptr[a + 1]; ++s)
      if (g_idx[s] < V) indices[t++] = (int32_t)(2 * n3 + g_idx[s]);
    for (int64_t s = indptr[r]; s < t; ++s) data[s] = 1.0 / (double)(1 + ((s * 2654435761u) & 1023));
  }
}

/* element residuals of a RANGE of cells with the gather U[cell_dofs] done here (the full-mesh leg does not hold [C][64]
 * copies of the states) and the scatter-add of Rl + Rn into F: one residual assembly as the port performs it. */
void fsi_c_assemble_residual(int64_t c0, int64_t c1, const double* coords, const int32_t* tets, const int32_t* tn, int64_t N2,
                             const int32_t* kind, const int32_t* region, const double* fluid_props, const double* solid_props,
                             const int32_t* solid_models, double dt, double theta, double delta, const double* N,
                             const double* dNref, const double* L, const double* qw, const double* U, const double* U1, double* F) {
  const OracleTables T = {N, dNref, L, qw};
#pragma omp parallel for schedule(static)
  for (int64_t c = c0; c < c1; ++c) {
    int64_t dofs[64];
    double xc[12], loc[64], loc1[64], rl[64], rn[64];
    for (int f = 0; f < 2; ++f)
      for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 10; ++a) dofs[30 * f + 10 * k + a] = f * 3 * N2 + 3 * (int64_t)tn[10 * c + a] + k;
    for (int a = 0; a < 4; ++a) dofs[60 + a] = 6 * N2 + tets[4 * c + a];
    for (int a = 0; a < 4; ++a)
      for (int k = 0; k < 3; ++k) xc[3 * a + k] = coords[3 * (int64_t)tets[4 * c + a] + k];
    for (int a = 0; a < 64; ++a) { loc[a] = U[dofs[a]]; loc1[a] = U1[dofs[a]]; }
    const int model = kind[c] == 1 ? solid_models[region[c]] : 0;
    r_element(xc, kind[c], cell_props(kind[c], region[c], fluid_props, solid_props), model, dt, theta, delta, &T, loc, loc1, rl, rn);
    for (int a = 0; a < 64; ++a) {
#pragma omp atomic
      F[dofs[a]] += rl[a] + rn[a];
    }
  }
}

/* element Jacobians (complex step, both parts) of a RANGE of cells with the same in-place gather; the 64 x 64 matrices are
 * reduced to a checksum per cell (sum of the entries) instead of being stored: the element arithmetic of a Jacobian
 * assembly without its scatter. */
void fsi_c_jacobian_elements(int64_t c0, int64_t c1, const double* coords, const int32_t* tets, const int32_t* tn, int64_t N2,
                             const int32_t* kind, const int32_t* region, const double* fluid_props, const double* solid_props,
                             const int32_t* solid_models, double dt, double theta, double delta, const double* N,
                             const double* dNref, const double* L, const double* qw, const double* U, const double* U1,
                             double* checksum) {
  const OracleTables T = {N, dNref, L, qw};
  const double h = 1e-30;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t c = c0; c < c1; ++c) {
    int64_t dofs[64];
    double xc[12], loc1[64];
    double complex z[64], rl[64], rn[64];
    for (int f = 0; f < 2; ++f)
      for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 10; ++a) dofs[30 * f + 10 * k + a] = f * 3 * N2 + 3 * (int64_t)tn[10 * c + a] + k;
    for (int a = 0; a < 4; ++a) dofs[60 + a] = 6 * N2 + tets[4 * c + a];
    for (int a = 0; a < 4; ++a)
      for (int k = 0; k < 3; ++k) xc[3 * a + k] = coords[3 * (int64_t)tets[4 * c + a] + k];
    for (int a = 0; a < 64; ++a) { z[a] = U[dofs[a]]; loc1[a] = U1[dofs[a]]; }
    const int model = kind[c] == 1 ? solid_models[region[c]] : 0;
    const double* props = cell_props(kind[c], region[c], fluid_props, solid_props);
    double acc = 0.0;
    for (int j = 0; j < 64; ++j) {
      const double complex keep = z[j];
      z[j] = keep + h * I;
      c_element(xc, kind[c], props, model, dt, theta, delta, &T, z, loc1, rl, rn);
      z[j] = keep;
      for (int a = 0; a < 64; ++a) acc += (cimag(rl[a]) + cimag(rn[a])) / h;
    }
    checksum[c - c0] = acc;
  }
}
