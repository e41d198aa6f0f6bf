"""ctypes binding of the C restatement of the oracle's element routines (oracle/fsi_oracle_c.c).

Test infrastructure, not product code: only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this.  ``oracle/fsi_oracle.py`` (numpy) stays the definition; this module is the same arithmetic
in C under OpenMP, checked against it to round-off in tests/test_oracle_c.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libfsi_oracle_c.so"
_lib = None


def build() -> Path:
    """gcc -O2 -fopenmp -shared (oracle/Makefile)."""
    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB


def load(build_if_missing: bool = True):
    global _lib
    if _lib is not None:
        return _lib
    src_newer = LIB.exists() and max((HERE / f).stat().st_mtime for f in ("fsi_oracle_c.c", "fsi_oracle_elem.inc")) > LIB.stat().st_mtime
    if (not LIB.exists() or src_newer) and build_if_missing:
        build()
    if not LIB.exists():
        return None
    lib = C.CDLL(str(LIB))
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    common = [i64, vp, vp, vp, vp, vp, vp, dbl, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.fsi_c_element_residuals.argtypes = common
    lib.fsi_c_element_residuals.restype = None
    lib.fsi_c_element_jacobians.argtypes = common
    lib.fsi_c_element_jacobians.restype = None
    lib.fsi_c_scatter_csr.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.fsi_c_scatter_csr.restype = i64
    lib.fsi_c_scatter_vector.argtypes = [i64, vp, vp, vp]
    lib.fsi_c_scatter_vector.restype = None
    lib.fsi_c_spmv.argtypes = [i64, vp, vp, vp, vp, vp]
    lib.fsi_c_spmv.restype = None
    lib.fsi_c_num_threads.restype = C.c_int
    full = [i64, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, dbl, dbl, dbl, vp, vp, vp, vp, vp, vp, vp]
    lib.fsi_c_assemble_residual.argtypes = full
    lib.fsi_c_assemble_residual.restype = None
    lib.fsi_c_jacobian_elements.argtypes = full
    lib.fsi_c_jacobian_elements.restype = None
    lib.fsi_c_node_graph.argtypes = [i64, vp, vp, vp, vp, vp, vp]
    lib.fsi_c_node_graph.restype = None
    lib.fsi_c_monolithic_pattern.argtypes = [i64, i64, vp, vp, vp, vp, vp]
    lib.fsi_c_monolithic_pattern.restype = None
    _lib = lib
    return lib


def available() -> bool:
    return os.environ.get("ORACLE_IMPL", "c") != "numpy" and load() is not None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CElements:
    """Element residuals / Jacobians of an ``FsiOracle`` through the C library (same inputs, same outputs)."""

    def __init__(self, o):
        from .fsi_oracle import DELTA
        self.o, self.lib = o, load()
        D = o.D
        self.xc = np.ascontiguousarray(o.x[o.tets], dtype=np.float64)                    # (C,4,3)
        self.kind = np.ascontiguousarray(o.kind, dtype=np.int32)
        self.region = np.ascontiguousarray(o.region, dtype=np.int32)
        self.fprops = np.ascontiguousarray(np.asarray(D["fluid_props"], dtype=np.float64).reshape(-1, 2))
        rows = [tuple(r) + (0.0,) * (6 - len(r)) for r in D["solid_props"]]
        self.sprops = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, 6))
        self.smodels = np.ascontiguousarray(D.get("solid_models", [0] * len(rows)), dtype=np.int32)
        self.N = np.ascontiguousarray(o.N)
        self.dN = np.ascontiguousarray(o.dNref)
        self.L = np.ascontiguousarray(o.L)
        self.qw = np.ascontiguousarray(o.qw)
        self.cell_dofs = np.ascontiguousarray(o.cell_dofs, dtype=np.int64)
        self.delta = float(DELTA)

    def _call(self, fn, loc, loc1, out_a, out_b):
        o = self.o
        fn(len(self.xc), _p(self.xc), _p(self.kind), _p(self.region), _p(self.fprops), _p(self.sprops), _p(self.smodels),
           o.dt, o.theta, self.delta, _p(self.N), _p(self.dN), _p(self.L), _p(self.qw), _p(loc), _p(loc1), _p(out_a), _p(out_b))

    def residuals(self, U, U1):
        loc = np.ascontiguousarray(U[self.cell_dofs]); loc1 = np.ascontiguousarray(U1[self.cell_dofs])
        Rl, Rn = np.empty_like(loc), np.empty_like(loc)
        self._call(self.lib.fsi_c_element_residuals, loc, loc1, Rl, Rn)
        return Rl, Rn

    def jacobians(self, U, U1):
        loc = np.ascontiguousarray(U[self.cell_dofs]); loc1 = np.ascontiguousarray(U1[self.cell_dofs])
        C_ = len(loc)
        Jl, Jn = np.empty((C_, 64, 64)), np.empty((C_, 64, 64))
        self._call(self.lib.fsi_c_element_jacobians, loc, loc1, Jl, Jn)
        return Jl, Jn

    # ---- assembly on a precomputed pattern ---------------------------------------------------------------
    def pattern(self):
        """CSR pattern of the monolithic matrix: every field couples over the P2 node graph (d, v) / the vertex graph (p),
        as DOLFIN's dofmap-based sparsity pattern does."""
        if getattr(self, "_pat", None) is None:
            import scipy.sparse as sp
            o = self.o
            tn, tv, N2, V = o.tn, o.tets, o.N2, o.V
            one = lambda r, c, shape: sp.csr_matrix((np.ones(r.size, dtype=np.int8), (r.ravel(), c.ravel())), shape=shape)
            G = one(np.repeat(tn[:, :, None], 10, 2), np.repeat(tn[:, None, :], 10, 1), (N2, N2))
            Gv = one(np.repeat(tn[:, :, None], 4, 2), np.repeat(tv[:, None, :], 10, 1), (N2, V))
            Gvv = one(np.repeat(tv[:, :, None], 4, 2), np.repeat(tv[:, None, :], 4, 1), (V, V))
            K = sp.kron(G, np.ones((3, 3), dtype=np.int8), format="csr")
            Kp = sp.kron(Gv, np.ones((3, 1), dtype=np.int8), format="csr")
            P = sp.bmat([[K, K, Kp], [K, K, Kp], [Kp.T, Kp.T, Gvv]], format="csr")
            P.sort_indices()
            self._pat = (P.indptr.astype(np.int64), P.indices.astype(np.int32))
        return self._pat

    def assemble_matrix(self, Je):
        import scipy.sparse as sp
        indptr, indices = self.pattern()
        data = np.zeros(len(indices))
        Je = np.ascontiguousarray(Je)
        missing = self.lib.fsi_c_scatter_csr(len(Je), _p(self.cell_dofs), _p(Je), _p(indptr), _p(indices), _p(data))
        if missing:
            raise RuntimeError(f"{missing} element entries outside the sparsity pattern")
        return sp.csr_matrix((data, indices, indptr), shape=(self.o.ndof, self.o.ndof))

    def assemble_vector(self, Re):
        F = np.zeros(self.o.ndof)
        Re = np.ascontiguousarray(Re)
        self.lib.fsi_c_scatter_vector(len(Re), _p(self.cell_dofs), _p(Re), _p(F))
        return F

    def spmv(self, A, x):
        y = np.empty(A.shape[0])
        ip = np.ascontiguousarray(A.indptr, dtype=np.int64); ix = np.ascontiguousarray(A.indices, dtype=np.int32)
        d = np.ascontiguousarray(A.data); x = np.ascontiguousarray(x)
        self.lib.fsi_c_spmv(A.shape[0], _p(ip), _p(ix), _p(d), _p(x), _p(y))
        return y
