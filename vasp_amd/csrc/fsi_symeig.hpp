// Eigen-decomposition of a small dense symmetric matrix on the host: Householder tridiagonalisation followed by the
// implicit QL iteration (the classical EISPACK pair tred2 / tql2, restated).  Used by the recycled-space compression of the
// Krylov solver (fsi_capi.hip: gcr_compress) on matrices of a few hundred rows; ~5 n^3 flops, 10-20 ms at n = 128.
// Plain C++ (no HIP): tests/test_symeig.py compiles it with g++ and checks it against LAPACK.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace fsi {

// A: n x n symmetric, row-major (only read).  On return V (row-major, n x n) holds the eigenvectors as COLUMNS and d the
// eigenvalues in ascending order:  A V[:, k] = d[k] V[:, k].
inline void sym_eig(int n, const std::vector<double>& A, std::vector<double>& V, std::vector<double>& d) {
  V = A;
  d.assign(n, 0.0);
  std::vector<double> e(n, 0.0);
  auto v = [&](int i, int j) -> double& { return V[(size_t)i * n + j]; };
  // ---- Householder reduction to tridiagonal form ----
  for (int j = 0; j < n; ++j) d[j] = v(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) { d[j] = v(i - 1, j); v(i, j) = 0.0; v(j, i) = 0.0; }
    } else {
      for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1], g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        v(j, i) = f;
        g = e[j] + v(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) { g += v(k, j) * d[k]; e[k] += v(k, j) * f; }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) v(k, j) -= (f * e[k] + g * d[k]);
        d[j] = v(i - 1, j);
        v(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  // accumulate the transformations
  for (int i = 0; i < n - 1; ++i) {
    v(n - 1, i) = v(i, i);
    v(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = v(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += v(k, i + 1) * v(k, j);
        for (int k = 0; k <= i; ++k) v(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) v(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) { d[j] = v(n - 1, j); v(n - 1, j) = 0.0; }
  v(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // ---- implicit QL on the tridiagonal matrix ----
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::pow(2.0, -52.0);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m >= n) m = n - 1;
    if (m > l) {
      int iter = 0;
      do {
        ++iter;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = v(k, i + 1);
            v(k, i + 1) = s * v(k, i) + c * h;
            v(k, i) = c * v(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 60);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  // ascending order
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j)
      if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; ++j) std::swap(v(j, i), v(j, k));
    }
  }
}

}  // namespace fsi
