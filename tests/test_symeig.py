"""Host-side numerics of the Krylov solver that can be checked without a GPU: the small symmetric eigen-solver the
recycled-space compression uses (vasp_amd/csrc/fsi_symeig.hpp: Householder tridiagonalisation + implicit QL) against LAPACK."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT

DRIVER = r"""
#include "fsi_symeig.hpp"
#include <cstdio>
int main() {
  int n;
  if (scanf("%d", &n) != 1) return 2;
  std::vector<double> A((size_t)n * n), V, d;
  for (auto& a : A) if (scanf("%lf", &a) != 1) return 2;
  fsi::sym_eig(n, A, V, d);
  for (int i = 0; i < n; ++i) printf("%.17g\n", d[i]);
  for (size_t i = 0; i < V.size(); ++i) printf("%.17g\n", V[i]);
  return 0;
}
"""


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("symeig")
    (d / "t.cpp").write_text(DRIVER)
    subprocess.run(["g++", "-O2", f"-I{ROOT / 'vasp_amd' / 'csrc'}", str(d / "t.cpp"), "-o", str(d / "t")], check=True)
    return d / "t"


@pytest.mark.parametrize("n,kind", [(1, "random"), (2, "random"), (9, "random"), (40, "clustered"), (96, "gram"), (60, "diagonal")])
def test_sym_eig_matches_lapack(driver, n, kind):
    rng = np.random.default_rng(n)
    if kind == "random":
        A = rng.standard_normal((n, n))
        A = A + A.T
    elif kind == "clustered":                       # many (nearly) equal eigenvalues: what D^T D of a well-preconditioned space looks like
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A = Q @ np.diag(np.r_[np.full(n - 6, 1.0) + 1e-9 * rng.standard_normal(n - 6), 1e-6, 1e-3, 0.1, 5.0, 40.0, 1e3]) @ Q.T
        A = 0.5 * (A + A.T)
    elif kind == "gram":                            # D^T D with D rank deficient
        D = rng.standard_normal((n // 3, n))
        A = D.T @ D
    else:
        A = np.diag(rng.standard_normal(n))
    text = f"{n}\n" + "\n".join(repr(float(x)) for x in A.ravel()) + "\n"
    out = subprocess.run([str(driver)], input=text, capture_output=True, text=True, check=True).stdout.split()
    d = np.array(out[:n], dtype=float)
    V = np.array(out[n:], dtype=float).reshape(n, n)
    scale = max(1.0, np.abs(A).max())
    assert np.all(np.diff(d) >= 0)                                         # ascending, as gcr_compress assumes
    assert np.allclose(d, np.linalg.eigvalsh(A), rtol=0, atol=1e-12 * scale * n)
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12 * n                   # orthonormal columns
    assert np.abs(A @ V - V * d).max() < 1e-12 * scale * n                 # A V[:, k] = d[k] V[:, k]
