"""Adapter that lets TESTS drive the host time loop (``vasp_amd.monolithic.run``) with the CPU oracle.

Test infrastructure only: the product's default backend is the HIP library and it never imports this.
"""
from __future__ import annotations

import numpy as np

from .fsi_oracle import FsiOracle


class OracleBackend:
    def __init__(self, desc):
        self.o = FsiOracle(desc)
        self.U = np.zeros(self.o.ndof)
        self.U1 = np.zeros(self.o.ndof)
        self.P = 0.0
        self.g = np.zeros(len(self.o.bc_dofs))
        self.o.solver_setup(self.U, self.U1)
        self.history = []

    def set_dirichlet_values(self, values):
        self.g = np.asarray(values, dtype=float).copy()

    def set_interface_pressure(self, P):
        self.P = float(P)

    def newton_solve(self, *, counter, first_step_num, atol, rtol, max_it, lmbda, recompute, recompute_tstep,
                     log=None):
        hist = self.o.newtonsolver(self.U, self.U1, self.P, self.g, atol=atol, rtol=rtol, max_it=max_it,
                                   lmbda=lmbda, recompute=recompute, recompute_tstep=recompute_tstep,
                                   counter=counter, first_step_num=first_step_num, log=log)
        self.history.append(hist)
        return hist

    def shift(self):
        self.U1[:] = self.U

    def get_state(self, which, out):
        out[:] = self.U if which == "n" else self.U1
        return out

    def set_state(self, which, x):
        (self.U if which == "n" else self.U1)[:] = x
