// Pointwise (quadrature-point) residual "fluxes" of the monolithic ALE-FSI form, generic over the scalar type.
//
// Restates, for one quadrature point, the integrands of turtleFSI's fluid_setup / solid_setup /
// extrapolate_setup as VaSP uses them:
//   value slots multiply the test function (psi: v-equation, phi: d-equation, gamma: p-equation),
//   gradient slots multiply its physical gradient.
// Instantiated with T = double for the residual kernel and T = Dual (forward-mode derivative along one
// local trial dof) for the Jacobian kernel; the Dual instantiation is what `derivative(F, dvp_["n"])` is in
// the reference.  The "linear"/"nonlinear" split mirrors F_*_linear / F_*_nonlinear there, because the
// reference assembles the Jacobian of the linear part once (A_pre) and of the nonlinear part at every refresh.
#pragma once
#include <hip/hip_runtime.h>

namespace fsi {

struct Dual {
  double v, e;
  __host__ __device__ Dual() : v(0.0), e(0.0) {}
  __host__ __device__ Dual(double a) : v(a), e(0.0) {}
  __host__ __device__ Dual(double a, double b) : v(a), e(b) {}
};
__host__ __device__ inline Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.e + b.e); }
__host__ __device__ inline Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.e - b.e); }
__host__ __device__ inline Dual operator-(Dual a) { return Dual(-a.v, -a.e); }
__host__ __device__ inline Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.v * b.e + a.e * b.v); }
__host__ __device__ inline Dual operator/(Dual a, Dual b) {
  double q = a.v / b.v;
  return Dual(q, (a.e - q * b.e) / b.v);
}
__host__ __device__ inline Dual operator+(Dual a, double b) { return Dual(a.v + b, a.e); }
__host__ __device__ inline Dual operator+(double a, Dual b) { return Dual(a + b.v, b.e); }
__host__ __device__ inline Dual operator-(Dual a, double b) { return Dual(a.v - b, a.e); }
__host__ __device__ inline Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.e); }
__host__ __device__ inline Dual operator*(Dual a, double b) { return Dual(a.v * b, a.e * b); }
__host__ __device__ inline Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.e); }
__host__ __device__ inline Dual operator/(Dual a, double b) { return Dual(a.v / b, a.e / b); }
__host__ __device__ inline Dual& operator+=(Dual& a, Dual b) { a.v += b.v; a.e += b.e; return a; }

__host__ __device__ inline double dlog(double a) { return log(a); }
__host__ __device__ inline Dual dlog(Dual a) { return Dual(log(a.v), a.e / a.v); }
__host__ __device__ inline double dpow(double a, double p) { return pow(a, p); }
__host__ __device__ inline Dual dpow(Dual a, double p) { const double q = pow(a.v, p - 1.0); return Dual(q * a.v, p * q * a.e); }

__host__ __device__ inline double value_of(double a) { return a; }
__host__ __device__ inline double value_of(Dual a) { return a.v; }
__host__ __device__ inline double deriv_of(double) { return 0.0; }
__host__ __device__ inline double deriv_of(Dual a) { return a.e; }

enum Part : int { PART_LINEAR = 1, PART_NONLINEAR = 2, PART_BOTH = 3 };

// State at one quadrature point: physical gradients g*[i][j] = d(field_i)/dx_j, values, pressure.
template <class T>
struct Kin {
  T gd[3][3], gv[3][3], d[3], v[3], p;
};
// What multiplies the test functions at this point (un-weighted).
template <class T>
struct Slots {
  T dval[3], dgrd[3][3];   // phi (d-equation)
  T vval[3], vgrd[3][3];   // psi (v-equation)
  T pval;                  // gamma (p-equation)
};

template <class T>
__host__ __device__ inline void zero_slots(Slots<T>& s) {
  for (int i = 0; i < 3; ++i) {
    s.dval[i] = T(0.0);
    s.vval[i] = T(0.0);
    for (int j = 0; j < 3; ++j) {
      s.dgrd[i][j] = T(0.0);
      s.vgrd[i][j] = T(0.0);
    }
  }
  s.pval = T(0.0);
}

// F = I + g; returns det F and inv F (cofactor form; analytic for Dual).
template <class T>
__host__ __device__ inline T inv_det_F(const T g[3][3], T Fi[3][3]) {
  T a = g[0][0] + 1.0, b = g[0][1], c = g[0][2];
  T d = g[1][0], e = g[1][1] + 1.0, f = g[1][2];
  T h = g[2][0], i = g[2][1], k = g[2][2] + 1.0;
  T c00 = e * k - f * i, c01 = c * i - b * k, c02 = b * f - c * e;
  T c10 = f * h - d * k, c11 = a * k - c * h, c12 = c * d - a * f;
  T c20 = d * i - e * h, c21 = b * h - a * i, c22 = a * e - b * d;
  T det = a * c00 + b * c10 + c * c20;
  T r = T(1.0) / det;
  Fi[0][0] = c00 * r; Fi[0][1] = c01 * r; Fi[0][2] = c02 * r;
  Fi[1][0] = c10 * r; Fi[1][1] = c11 * r; Fi[1][2] = c12 * r;
  Fi[2][0] = c20 * r; Fi[2][1] = c21 * r; Fi[2][2] = c22 * r;
  return det;
}

struct FluidProps { double rho, mu; };
struct SolidProps { double rho, mu, lam; int model; double C10, C01, C11; };   // model: 0 StVenantKirchoff, 1 MooneyRivlin
struct Scheme { double k, th0, th1, delta, alpha; };

// turtleFSI fluid.py (+ laplace.py "constant") on a fluid cell.  `o` is the state at n-1 (plain doubles).
template <class T, int PART>
__host__ __device__ inline void fluid_flux(const FluidProps& fp, const Scheme& sc, const Kin<T>& s,
                                           const Kin<double>& o, Slots<T>& out) {
  const double rho = fp.rho, mu = fp.mu, rk = fp.rho / sc.k;
  zero_slots(out);
  // the n-1 part first: the gradients of the old state are dead before the nonlinear part needs its registers
  if (PART & PART_LINEAR) {
    double Fi1[3][3];
    double J1 = inv_det_F(o.gd, Fi1);
    double A1[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A1[i][j] = o.gv[i][0] * Fi1[0][j] + o.gv[i][1] * Fi1[1][j] + o.gv[i][2] * Fi1[2][j];
    for (int i = 0; i < 3; ++i) {
      double conv1 = A1[i][0] * o.v[0] + A1[i][1] * o.v[1] + A1[i][2] * o.v[2];
      out.vval[i] = (rk * J1 * sc.th1) * (s.v[i] - o.v[i]) + sc.th1 * rho * J1 * conv1;
    }
    double sg1[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) sg1[i][j] = (sc.th1 * mu) * (A1[i][j] + A1[j][i]);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        out.vgrd[i][j] = T(J1 * (sg1[i][0] * Fi1[j][0] + sg1[i][1] * Fi1[j][1] + sg1[i][2] * Fi1[j][2]));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.dgrd[i][j] = sc.alpha * s.gd[i][j];
  }
  if (PART & PART_NONLINEAR) {
    T Fi[3][3];
    T J = inv_det_F(s.gd, Fi);
    T A[3][3];                                   // grad(v) * inv(F)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[i][j] = s.gv[i][0] * Fi[0][j] + s.gv[i][1] * Fi[1][j] + s.gv[i][2] * Fi[2][j];
    T w[3];                                      // theta0 v - (d - d1)/k : convecting velocity incl. ALE term
    for (int j = 0; j < 3; ++j) w[j] = sc.th0 * s.v[j] - (s.d[j] - o.d[j]) * (1.0 / sc.k);
    for (int i = 0; i < 3; ++i) {
      T conv = A[i][0] * w[0] + A[i][1] * w[1] + A[i][2] * w[2];
      out.vval[i] = J * (rk * sc.th0 * (s.v[i] - o.v[i]) + rho * conv) + out.vval[i];
    }
    // J (-p I + theta0 mu (A + A^T)) F^-T
    T sg[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) sg[i][j] = (sc.th0 * mu) * (A[i][j] + A[j][i]);
    for (int i = 0; i < 3; ++i) sg[i][i] = sg[i][i] - s.p;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        out.vgrd[i][j] = J * (sg[i][0] * Fi[j][0] + sg[i][1] * Fi[j][1] + sg[i][2] * Fi[j][2]) + out.vgrd[i][j];
    out.pval = J * (A[0][0] + A[1][1] + A[2][2]);  // div(J F^-1 v) = J tr(grad(v) F^-1)  (Piola identity)
  }
}

// First Piola-Kirchhoff stress of the St. Venant-Kirchhoff model: F (lambda tr(E) I + 2 mu E).
template <class T>
__host__ __device__ inline void piola_svk(const SolidProps& sp, const T g[3][3], T P[3][3]) {
  T F[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) F[i][j] = g[i][j] + (i == j ? 1.0 : 0.0);
  T E[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j) {
      T c = F[0][i] * F[0][j] + F[1][i] * F[1][j] + F[2][i] * F[2][j];
      E[i][j] = 0.5 * (c - (i == j ? 1.0 : 0.0));
      E[j][i] = E[i][j];
    }
  T tr = E[0][0] + E[1][1] + E[2][2];
  T S[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[i][j] = (2.0 * sp.mu) * E[i][j];
  for (int i = 0; i < 3; ++i) S[i][i] = S[i][i] + sp.lam * tr;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) P[i][j] = F[i][0] * S[0][j] + F[i][1] * S[1][j] + F[i][2] * S[2][j];
}

// First Piola-Kirchhoff stress of a compressible Mooney-Rivlin solid (material_model "MooneyRivlin" with C10, C01, C11
// [REF src/vasp/simulations/avf.py:77-80, predeform.py:71-72]):
//   psi = C10 (I1b - 3) + C01 (I2b - 3) + C11 (I1b - 3)(I2b - 3) + K (J ln J - J + 1),   K = lambda + 2 mu / 3,
// with the isochoric invariants I1b = J^-2/3 tr C, I2b = J^-4/3 (tr(C)^2 - tr(C^2))/2;  S = 2 dpsi/dC, P = F S.
// The reference takes S from turtleFSI by automatic differentiation of its strain energy; that source is not in the
// tree and the reference's tests pin no number for this model: parity unpinned.
template <class T>
__host__ __device__ inline void piola_mr(const SolidProps& sp, const T g[3][3], T P[3][3]) {
  T F[3][3], C[3][3], Ci[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) F[i][j] = g[i][j] + (i == j ? 1.0 : 0.0);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = F[0][i] * F[0][j] + F[1][i] * F[1][j] + F[2][i] * F[2][j];
  T Cm[3][3];                                      // C - I, so that inv_det_F (which inverts I + g) applies
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Cm[i][j] = C[i][j] - (i == j ? 1.0 : 0.0);
  const T detC = inv_det_F(Cm, Ci);
  T Fi[3][3];
  const T J = inv_det_F(g, Fi);
  (void)detC;
  const T I1 = C[0][0] + C[1][1] + C[2][2];
  T trC2 = T(0.0);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) trC2 = trC2 + C[i][j] * C[j][i];
  const T I2 = 0.5 * (I1 * I1 - trC2);
  const T Jm23 = dpow(J, -2.0 / 3.0), Jm43 = Jm23 * Jm23;
  const T I1b = Jm23 * I1, I2b = Jm43 * I2;
  const T a1 = 2.0 * (sp.C10 + sp.C11 * (I2b - 3.0)) * Jm23;
  const T a2 = 2.0 * (sp.C01 + sp.C11 * (I1b - 3.0)) * Jm43;
  const double K = sp.lam + 2.0 * sp.mu / 3.0;
  const T vol = K * dlog(J) * J;
  T S[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double dij = (i == j) ? 1.0 : 0.0;
      S[i][j] = a1 * (dij - (I1 * (1.0 / 3.0)) * Ci[i][j]) + a2 * (I1 * dij - C[i][j] - (I2 * (2.0 / 3.0)) * Ci[i][j]) +
                vol * Ci[i][j];
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) P[i][j] = F[i][0] * S[0][j] + F[i][1] * S[1][j] + F[i][2] * S[2][j];
}
template <class T>
__host__ __device__ inline void piola(const SolidProps& sp, const T g[3][3], T P[3][3]) {
  if (sp.model == 1) piola_mr<T>(sp, g, P);
  else piola_svk<T>(sp, g, P);
}

// turtleFSI solid.py on a solid cell.
template <class T, int PART>
__host__ __device__ inline void solid_flux(const SolidProps& sp, const Scheme& sc, const Kin<T>& s,
                                           const Kin<double>& o, Slots<T>& out) {
  zero_slots(out);
  const double rk = sp.rho / sc.k;
  if (PART & PART_LINEAR) {
    for (int i = 0; i < 3; ++i) {
      out.vval[i] = rk * (s.v[i] - o.v[i]);
      out.dval[i] = (sc.delta * rk) * (s.d[i] - o.d[i]) - (sc.delta * sp.rho) * (sc.th0 * s.v[i] + sc.th1 * o.v[i]);
    }
    double P1[3][3];
    piola<double>(sp, o.gd, P1);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.vgrd[i][j] = T(sc.th1 * P1[i][j]);
  }
  if (PART & PART_NONLINEAR) {
    T P[3][3];
    piola<T>(sp, s.gd, P);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.vgrd[i][j] = out.vgrd[i][j] + sc.th0 * P[i][j];
  }
}

}  // namespace fsi
