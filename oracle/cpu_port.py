"""CPU baseline of bench.py: the CPU port of the algorithm, timed on the host cores on a bounded sample of the bench
workload.

Test infrastructure (only ``bench.py``'s ``cpu_baseline`` leg calls this; nothing here is measured as the product).
What runs: the offset-stenosis problem file on a mesh of the bench's own generator (``vasp_amd.meshgen``), the reference's
quasi-Newton policy (``oracle.fsi_oracle.FsiOracle.newtonsolver``) with
  * residual and Jacobian assembly by the C restatement under OpenMP on all host cores (oracle/fsi_oracle_c.c), and
  * the linear solve by an exact sparse LU re-used across Newton iterations - what the reference itself does with MUMPS
    [REF src/vasp/simulations/offset_stenosis.py:44-48] (SuperLU here, one thread).
The sample is sized so that set-up + time steps take about the budget; throughput is reported in the bench's unit and,
because the sample is ~150x smaller than the bench mesh, also per degree of freedom.
"""
from __future__ import annotations

import contextlib
import io
import os
import tempfile
import time
from pathlib import Path

import numpy as np


def timed_newton_run(budget_s: float = 25.0, tets: int = 6500, steps: int = 20) -> dict:
    """One Jacobian lifetime of the reference's policy (``recompute_tstep = 20`` [REF offset_stenosis.py:44-48]): the first
    step assembles the Jacobian and factorises it, the other 19 re-use the factors - timed separately, because the
    reference amortises its factorisation exactly this way and a one-step sample understates the CPU path by an order of
    magnitude.  ``budget_s`` only bounds the steady steps: the refresh step takes what it takes."""
    from vasp_amd.meshgen import write_mesh
    from vasp_amd.monolithic import prepare
    from .backend import OracleBackend
    from . import c_oracle

    tmp = Path(tempfile.mkdtemp(prefix="vaspfsi_cpu_baseline_"))
    write_mesh(tmp / "stenosis.h5", tets, seed=0)
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(
            ["-p", "offset_stenosis", "-dt", "0.001", "-T", "1.0", "--theta", "0.501", "--verbose", "False", "--folder",
             str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={tmp / 'stenosis.h5'}"])
    t0 = time.perf_counter()
    ob = OracleBackend(desc)            # includes A_pre = assemble(J_linear), as the reference's solver_setup
    kind = "port"
    its, step, t = 0, 0, 0.0
    t_first, its_first = None, 0
    t_steady0 = None
    while step < steps and (step < 2 or time.perf_counter() - t_steady0 < max(budget_s, 1.0)):
        t += 0.001
        with contextlib.redirect_stdout(io.StringIO()):
            ns["t"] = t
            hook("pre_solve")(**ns)
        ob.set_dirichlet_values(bc_values())
        ob.set_interface_pressure(float(pressure.P))
        hist = ob.newton_solve(counter=step, first_step_num=0, atol=1e-6, rtol=1e-6, max_it=50, lmbda=1.0, recompute=20,
                               recompute_tstep=20)
        ob.shift()
        its += len(hist)
        step += 1
        if t_first is None:
            t_first, its_first = time.perf_counter() - t0, len(hist)
            t_steady0 = time.perf_counter()
    dt = time.perf_counter() - t0
    o = ob.o
    cores = c_oracle.load().fsi_c_num_threads() if o.c is not None else 1
    # kernel rates of the port on the same sample (all cores): residual assembly and the sparse product
    U, U1 = ob.U.copy(), ob.U1.copy()
    ta = time.perf_counter()
    reps = 0
    while time.perf_counter() - ta < 1.0:
        o.residual(U, U1, 0.0)
        reps += 1
    res_rate = reps * len(o.tets) / (time.perf_counter() - ta)
    spmv_gbs = None
    if o.c is not None and getattr(o, "A", None) is not None:
        A = o.A.tocsr()
        x = np.ones(A.shape[0])
        tb = time.perf_counter()
        reps = 0
        while time.perf_counter() - tb < 1.0:
            o.c.spmv(A, x)
            reps += 1
        spmv_gbs = reps * (A.nnz * 12.0 + A.shape[0] * 20.0) / (time.perf_counter() - tb) / 1e9
    steady_s = dt - t_first
    steady_its, steady_steps = its - its_first, step - 1
    steady_rate = steady_its / steady_s if steady_steps > 0 and steady_s > 0 else None
    # one Jacobian lifetime = the refresh step + (steps - 1) steady steps; if the budget cut the run short the remaining
    # steady steps are priced at the measured steady rate
    life_s = t_first + (steady_s / steady_steps * (steps - 1) if steady_steps > 0 else 0.0)
    life_its = its_first + (steady_its / steady_steps * (steps - 1) if steady_steps > 0 else 0.0)
    return {"value": life_its / life_s, "unit": "Newton-iterations/s", "cores": int(cores), "kind": kind,
            "sample": f"offset_stenosis on a {len(o.tets)}-tet mesh of the bench's generator ({o.ndof} dofs), dt=1e-3, theta=0.501, "
                      f"quasi-Newton atol=rtol=1e-6 recompute_tstep=20, ONE Jacobian lifetime of {steps} time steps: the refresh step "
                      f"(Jacobian assembly + sparse LU) {t_first:.1f} s for {its_first} Newton iterations, then {steady_steps} steady "
                      f"steps re-using the factors, {steady_its} Newton iterations in {steady_s:.1f} s; `value` = iterations of the "
                      f"lifetime / its time.  Assembly in C/OpenMP on {cores} threads, LU and triangular solves SuperLU on 1 thread "
                      f"(the reference's linear solver is a direct LU, MUMPS).  The generator's next mesh size is 48 000 tets, "
                      f"whose serial LU alone takes tens of minutes: the 6 000-tet sample is what fits a bench run",
            "steady_state": {"value": steady_rate, "unit": "Newton-iterations/s", "steps": steady_steps, "newton_iterations": steady_its,
                             "seconds": steady_s},
            "refresh_step": {"seconds": t_first, "newton_iterations": its_first},
            "lifetime_steps": steps, "tets": int(len(o.tets)), "dofs": int(o.ndof),
            "dof_updates_per_s": life_its / life_s * o.ndof,
            "residual_assembly_tets_per_s": res_rate, "spmv_GBps": spmv_gbs, "host_cpus": os.cpu_count()}


# ---------------------------------------------------------------------------------------------------------------------
# CPU Krylov leg (round 4; VERDICT r3 missing 6 / item 5c): "assembly + SpMV + Krylov" on a bench-like mesh size
# ---------------------------------------------------------------------------------------------------------------------
class CpuKrylov:
    """The linear solver of the HIP path restated for the host: right-preconditioned GCR whose directions are kept while the
    Jacobian is kept, on the row-equilibrated monolithic matrix, with the same approximate block factorisation as
    vasp_amd/csrc/fsi_block.hip - solid displacement eliminated (d = k theta v on solid nodes), velocity predictor solid ->
    fluid, SIMPLE-type Schur complement for the pressure, velocity correction, then the displacement block - and, where the
    GPU runs fixed numbers of Chebyshev sweeps, scipy's incomplete LU of each block (spilu: what a CPU is good at - few,
    serial, strong inner solves).  Assembly and the monolithic product run in C under OpenMP on all cores (oracle/fsi_oracle_c.c)."""

    def __init__(self, o):
        import scipy.sparse as sp
        self.o, self.sp = o, sp
        N2, V = o.N2, o.V
        self.N2, self.V = N2, V
        solid_nodes = np.zeros(N2, dtype=bool)
        solid_nodes[o.tn[o.kind == 1].ravel()] = True
        self.solid3 = np.repeat(solid_nodes, 3)                      # velocity dofs of solid (incl. interface) nodes
        self.cap, self.m = 400, 0
        self.Qs = self.Ps = None                                     # kept directions: rows of (cap, n) arrays, Q = A P orthonormal
        self.applies = 0

    def refresh(self, A):
        """New Jacobian: equilibrate, split into field blocks, factorise the blocks incompletely, drop the kept directions."""
        import scipy.sparse.linalg as spla
        sp, N2, V, o = self.sp, self.N2, self.V, self.o
        n3 = 3 * N2
        A = A.tocsr()
        rs = 1.0 / np.maximum(np.abs(A).max(axis=1).toarray().ravel(), 1e-300)
        self.rs = rs
        self.A = (sp.diags(rs) @ A).tocsr()
        A = self.A
        d, v, p = slice(0, n3), slice(n3, 2 * n3), slice(2 * n3, 2 * n3 + V)
        kth = o.dt * o.theta
        S3 = sp.diags(self.solid3.astype(float))
        Add, Adv = A[d, d].tocsr(), A[d, v].tocsr()
        Avv = (A[v, v] + kth * (A[v, d] @ S3)).tocsr()               # solid d eliminated: delta d = k theta delta v there
        Apv = (A[p, v] + kth * (A[p, d] @ S3)).tocsr()
        Avp, App = A[v, p].tocsr(), A[p, p].tocsr()
        Dinv = 1.0 / Avv.diagonal()
        Sc = (App - Apv @ sp.diags(Dinv) @ Avp).tocsc()
        ilu = lambda M: spla.spilu(M.tocsc(), drop_tol=1e-3, fill_factor=6.0)
        s_idx, f_idx = np.nonzero(self.solid3)[0], np.nonzero(~self.solid3)[0]
        self.s_idx, self.f_idx = s_idx, f_idx
        self.Mss = ilu(Avv[s_idx][:, s_idx])                         # solid block: thin-walled elasticity, fully coupled, small
        self.Afs = Avv[f_idx][:, s_idx].tocsr()
        # fluid-interior velocity block: its component-diagonal part (what the GPU sweeps run on), one factorisation per
        # component; displacement block = A0 (x) I3 (solid mass / mesh Laplacian): one scalar factorisation for all three
        self.f_comp = [f_idx[f_idx % 3 == c] for c in range(3)]
        self.Mff = [ilu(Avv[fc][:, fc]) for fc in self.f_comp]
        self.f_pos = [np.nonzero(f_idx % 3 == c)[0] for c in range(3)]
        self.Msc = ilu(Sc)
        self.Mdd = ilu(Add[0::3][:, 0::3])
        self.Adv, self.Apv, self.Avp, self.Dinv = Adv, Apv, Avp, Dinv
        if self.Qs is None:
            self.Qs, self.Ps = np.empty((self.cap, A.shape[0])), np.empty((self.cap, A.shape[0]))
        self.m = 0

    def precondition(self, r):
        N2, V = self.N2, self.V
        n3 = 3 * N2
        rd, rv, rp = r[:n3], r[n3:2 * n3], r[2 * n3:]
        vs = np.zeros(n3)
        xs = self.Mss.solve(rv[self.s_idx])
        vs[self.s_idx] = xs
        rf = rv[self.f_idx] - self.Afs @ xs
        for c in range(3):
            vs[self.f_comp[c]] = self.Mff[c].solve(rf[self.f_pos[c]])
        dp = self.Msc.solve(rp - self.Apv @ vs)
        dv = vs - self.Dinv * (self.Avp @ dp)
        td = rd - self.Adv @ dv
        dd = np.empty(n3)
        for c in range(3):
            dd[c::3] = self.Mdd.solve(np.ascontiguousarray(td[c::3]))
        self.applies += 1
        return np.concatenate([dd, dv, dp])

    def spmv(self, x):
        o = self.o
        return o.c.spmv(self.A, x) if o.c is not None else self.A @ x

    def solve(self, b, rtol, max_it=400):
        """A du = b (un-equilibrated rhs in, solution out); returns (du, iterations, achieved relative residual)."""
        bs = self.rs * b
        bn = float(np.linalg.norm(bs))
        x = np.zeros_like(bs)
        if bn == 0.0:
            return x, 0, 0.0
        r = bs.copy()
        Q, Pd = self.Qs, self.Ps
        if self.m > 0:                                               # projection on the recycled space
            h = Q[:self.m] @ r
            x += h @ Pd[:self.m]
            r -= h @ Q[:self.m]
        its = 0
        rn = float(np.linalg.norm(r))
        while rn > rtol * bn and its < max_it:
            z = self.precondition(r)
            w = self.spmv(z)
            m = self.m
            if m > 0:
                for _ in range(2):                                   # classical Gram-Schmidt, twice
                    h = Q[:m] @ w
                    w -= h @ Q[:m]
                    z -= h @ Pd[:m]
            wn = float(np.linalg.norm(w))
            if not np.isfinite(wn) or wn == 0.0:
                break
            q, pdir = w / wn, z / wn
            a = float(q @ r)
            x += a * pdir
            r -= a * q
            slot = m if m < self.cap else self.cap - 64 + (its % 64)   # full store: the oldest directions stay, a ring of 64 rotates
            Q[slot], Pd[slot] = q, pdir
            self.m = min(m + 1, self.cap)
            its += 1
            rn = float(np.linalg.norm(r))
        return x, its, rn / bn


def timed_krylov_run(budget_s: float = 30.0, tets: int = 48000, max_steps: int = 20) -> dict:
    """The reference's quasi-Newton loop on the offset-stenosis problem with a mesh of the bench's generator at ``tets``
    cells, linear solves by ``CpuKrylov`` with the production forcing term (1e-2 atol / |b|, as fsi_newton_solve): what
    SURVEY.md section 8(d) asks of config 2's CPU leg - assembly + SpMV + Krylov on the host cores.  The refresh step (Jacobian
    assembly, block factorisations) is timed separately from the steady steps that follow, and the run stops after ``budget_s``
    of steady stepping; `value` prices one Jacobian lifetime of ``max_steps`` steps as the direct-LU baseline above does."""
    from vasp_amd.meshgen import write_mesh
    from vasp_amd.monolithic import prepare
    from . import c_oracle
    from .fsi_oracle import FsiOracle

    tmp = Path(tempfile.mkdtemp(prefix="vaspfsi_cpu_krylov_"))
    write_mesh(tmp / "stenosis.h5", tets, seed=0)
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(
            ["-p", "offset_stenosis", "-dt", "0.001", "-T", "1.0", "--theta", "0.501", "--verbose", "False", "--folder",
             str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={tmp / 'stenosis.h5'}"])
    t_all = time.perf_counter()
    o = FsiOracle(desc)
    U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
    o.solver_setup(U, U1)
    ks = CpuKrylov(o)
    t_setup = time.perf_counter() - t_all
    atol = rtol = 1e-6
    phase = {"jacobian_s": 0.0, "factor_s": 0.0, "residual_s": 0.0, "krylov_s": 0.0}
    its_total, kry_total, step, t = 0, 0, 0, 0.0
    t_first, its_first = None, 0
    t0 = time.perf_counter()
    t_steady0 = None
    while step < max_steps and (step < 1 or time.perf_counter() - t_steady0 < budget_s):
        t += 0.001
        with contextlib.redirect_stdout(io.StringIO()):
            ns["t"] = t
            hook("pre_solve")(**ns)
        g, P = bc_values(), float(pressure.P)
        it, residual, rel_res, last = 0, 1e8, 1e8, 1e8
        while rel_res > rtol and residual > atol and it < 50:
            rec = (it == 0 and step % 20 == 0) or (it > 0 and it % 20 == 0) or (it > 0 and last < residual)
            if rec:
                ta = time.perf_counter()
                A = o.jacobian(U, U1)
                tb = time.perf_counter()
                ks.refresh(A)
                phase["jacobian_s"] += tb - ta
                phase["factor_s"] += time.perf_counter() - tb
            ta = time.perf_counter()
            b = o.rhs(U, U1, P, g)
            phase["residual_s"] += time.perf_counter() - ta
            bn = float(np.linalg.norm(b))
            last = residual
            eta = max(1e-10, min(1e-2, 1e-2 * atol / bn)) if bn > 0 else 1e-10
            ta = time.perf_counter()
            dU, kits, rr = ks.solve(b, eta)
            phase["krylov_s"] += time.perf_counter() - ta
            kry_total += kits
            U += dU
            U[o.bc_dofs] = g
            residual, rel_res = bn, o.function_norm(dU)
            it += 1
            if not np.isfinite(residual) or not np.isfinite(rel_res) or residual > 1e20:
                raise RuntimeError("the CPU Krylov baseline diverged")
        U1[:] = U
        its_total += it
        step += 1
        if t_first is None:
            t_first, its_first = time.perf_counter() - t0, it
            t_steady0 = time.perf_counter()
    dt = time.perf_counter() - t0
    cores = c_oracle.load().fsi_c_num_threads() if o.c is not None else 1
    steady_s, steady_its, steady_steps = dt - t_first, its_total - its_first, step - 1
    life_s = t_first + (steady_s / steady_steps * (max_steps - 1) if steady_steps > 0 else 0.0)
    life_its = its_first + (steady_its / steady_steps * (max_steps - 1) if steady_steps > 0 else 0.0)
    return {"value": life_its / life_s, "unit": "Newton-iterations/s", "cores": int(cores), "kind": "port",
            "sample": f"offset_stenosis on a {len(o.tets)}-tet mesh of the bench's generator ({o.ndof} dofs), dt=1e-3, theta=0.501, "
                      f"quasi-Newton atol=rtol=1e-6 recompute_tstep=20, forcing 1e-2: the refresh step ({its_first} Newton iterations: Jacobian "
                      f"assembly + incomplete factorisation of the field blocks) took {t_first:.1f} s, then {steady_steps} steady step(s) with "
                      f"{steady_its} Newton iterations in {steady_s:.1f} s; {kry_total} GCR iterations in all; `value` = Newton iterations of "
                      f"one Jacobian lifetime of {max_steps} steps / its time (steady steps beyond the budget priced at the measured rate).  "
                      f"Assembly and the monolithic SpMV in C/OpenMP on {cores} threads; field-split preconditioner of the HIP path with "
                      f"scipy spilu inner solves (serial) instead of Chebyshev sweeps; GCR with recycled directions (numpy BLAS)",
            "steady_state": {"value": (steady_its / steady_s) if steady_steps > 0 and steady_s > 0 else None, "unit": "Newton-iterations/s",
                             "steps": steady_steps, "newton_iterations": steady_its, "seconds": steady_s},
            "refresh_step": {"seconds": t_first, "newton_iterations": its_first},
            "phase_s": phase, "setup_s": t_setup, "krylov_iterations": kry_total, "preconditioner_applications": ks.applies,
            "lifetime_steps": max_steps, "tets": int(len(o.tets)), "dofs": int(o.ndof),
            "dof_updates_per_s": life_its / life_s * o.ndof, "host_cpus": os.cpu_count()}


# ---------------------------------------------------------------------------------------------------------------------
# Kernel legs on the bench mesh ITSELF (round 5): residual assembly, Jacobian element arithmetic and the
# monolithic SpMV of the CPU port at config 2's size, all host cores, seconds
# ---------------------------------------------------------------------------------------------------------------------
class FullMeshKernels:
    """The C/OpenMP element routines and the CSR product on a mesh of any size without FsiOracle's per-cell tables (its
    physical-gradient array alone is 6.5 GB at 1.12 M tets): states are gathered cell by cell inside the C loops
    (oracle/fsi_oracle_c.c: fsi_c_assemble_residual / fsi_c_jacobian_elements), and the monolithic matrix's sparsity pattern is
    built from the P2 node graph in C (fsi_c_node_graph / fsi_c_monolithic_pattern) - the same pattern ``CElements.pattern``
    builds with scipy for the small cases (tests/test_oracle_c.py holds the two together)."""

    def __init__(self, desc):
        from . import c_oracle
        from .fsi_oracle import DELTA, keast24, tabulate_p2
        self.lib = c_oracle.load()
        if self.lib is None:
            raise RuntimeError("oracle/libfsi_oracle_c.so is missing (make -C oracle)")
        self._p = c_oracle._p
        self.coords = np.ascontiguousarray(desc["coords"], dtype=np.float64)
        self.tets = np.ascontiguousarray(desc["tets"], dtype=np.int32)
        self.tn = np.ascontiguousarray(desc["tet_nodes"], dtype=np.int32)
        self.N2, self.V, self.C = int(desc["num_nodes"]), len(self.coords), len(self.tets)
        self.ndof = 6 * self.N2 + self.V
        self.kind = np.ascontiguousarray(desc["cell_kind"], dtype=np.int32)
        self.region = np.ascontiguousarray(desc["cell_region"], dtype=np.int32)
        self.fprops = np.ascontiguousarray(np.asarray(desc["fluid_props"], dtype=np.float64).reshape(-1, 2))
        rows = [tuple(r) + (0.0,) * (6 - len(r)) for r in desc["solid_props"]]
        self.sprops = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, 6))
        self.smodels = np.ascontiguousarray(desc.get("solid_models", [0] * len(rows)), dtype=np.int32)
        self.dt, self.theta, self.delta = float(desc["dt"]), float(desc["theta"]), float(DELTA)
        qp, qw = keast24()
        N, dN, L, _ = tabulate_p2(qp)
        self.tab = [np.ascontiguousarray(a) for a in (N, dN, L, qw)]

    def _args(self, c0, c1, U, U1, out):
        p = self._p
        return (c0, c1, p(self.coords), p(self.tets), p(self.tn), self.N2, p(self.kind), p(self.region), p(self.fprops), p(self.sprops),
                p(self.smodels), self.dt, self.theta, self.delta, *[p(a) for a in self.tab], p(U), p(U1), p(out))

    def assemble_residual(self, U, U1, F=None, cells=None):
        """F += sum over the cells [c0, c1) of (Rl + Rn) scattered to the dofs: the volume part of assemble(F)."""
        c0, c1 = cells if cells is not None else (0, self.C)
        F = np.zeros(self.ndof) if F is None else F
        self.lib.fsi_c_assemble_residual(*self._args(c0, c1, U, U1, F))
        return F

    def jacobian_elements(self, U, U1, c0, c1):
        """Complex-step element Jacobians (both parts of the split) of the cells [c0, c1); returns one checksum per cell."""
        out = np.empty(c1 - c0)
        self.lib.fsi_c_jacobian_elements(*self._args(c0, c1, U, U1, out))
        return out

    def pattern(self):
        """(indptr, indices, data) of the monolithic CSR matrix on the mesh's own sparsity pattern; ``data`` holds deterministic
        stand-in values (the product's time does not depend on them), first touched by the threads that stream them."""
        p = self._p
        flat = self.tn.ravel()
        order = np.argsort(flat, kind="stable")
        inc = (order // 10).astype(np.int32)
        inc_ptr = np.zeros(self.N2 + 1, dtype=np.int64)
        np.cumsum(np.bincount(flat, minlength=self.N2), out=inc_ptr[1:])
        deg = np.empty(self.N2, dtype=np.int32)
        self.lib.fsi_c_node_graph(self.N2, p(self.tn), p(inc_ptr), p(inc), None, None, p(deg))
        g_ptr = np.zeros(self.N2 + 1, dtype=np.int64)
        np.cumsum(deg, out=g_ptr[1:])
        g_idx = np.empty(int(g_ptr[-1]), dtype=np.int32)
        self.lib.fsi_c_node_graph(self.N2, p(self.tn), p(inc_ptr), p(inc), p(g_ptr), p(g_idx), None)
        degv = np.add.reduceat((g_idx < self.V).astype(np.int64), g_ptr[:-1])          # vertex neighbours of every node
        rowlen_node = 6 * deg.astype(np.int64) + degv
        rowlen = np.concatenate([np.repeat(rowlen_node, 3), np.repeat(rowlen_node, 3), rowlen_node[:self.V]])
        indptr = np.zeros(self.ndof + 1, dtype=np.int64)
        np.cumsum(rowlen, out=indptr[1:])
        nnz = int(indptr[-1])
        indices, data = np.empty(nnz, dtype=np.int32), np.empty(nnz)
        self.lib.fsi_c_monolithic_pattern(self.N2, self.V, p(g_ptr), p(g_idx), p(indptr), p(indices), p(data))
        return indptr, indices, data

    def spmv(self, indptr, indices, data, x, y):
        p = self._p
        self.lib.fsi_c_spmv(self.ndof, p(indptr), p(indices), p(data), p(x), p(y))
        return y


def full_mesh_kernels(desc, budget_s: float = 12.0, max_nnz: float = 2.5e9) -> dict:
    """Residual assembly (whole mesh), Jacobian element arithmetic (a bounded run of consecutive cells of the same mesh) and the
    monolithic CSR product (whole matrix pattern) of the CPU port on ``desc``'s mesh - bench.py passes the 1.12 M-tet bench mesh.
    Rates in the units the GPU kernels are reported in: tets/s and GB/s of algorithmic bytes (12 B per entry +
    20 B per row for CSR f64 + i32)."""
    t_all = time.perf_counter()
    k = FullMeshKernels(desc)
    cores = k.lib.fsi_c_num_threads()
    rng = np.random.default_rng(0)
    h = float(np.abs(k.coords[k.tets[0, 1]] - k.coords[k.tets[0, 0]]).max())
    U, U1 = np.zeros(k.ndof), np.zeros(k.ndof)                       # a small smooth-ish state: no cell is degenerate, no term vanishes
    U[:3 * k.N2] = 1e-3 * h * rng.standard_normal(3 * k.N2)
    U1[:3 * k.N2] = 0.9 * U[:3 * k.N2]
    U[3 * k.N2:6 * k.N2] = 0.1 * rng.standard_normal(3 * k.N2)
    U1[3 * k.N2:6 * k.N2] = 0.9 * U[3 * k.N2:6 * k.N2]
    U[6 * k.N2:] = rng.standard_normal(k.V)
    out = {"cores": int(cores), "tets": int(k.C), "dofs": int(k.ndof), "host_cpus": os.cpu_count()}
    # residual assembly: whole mesh, repeated while the budget lasts (at least once)
    F = np.zeros(k.ndof)
    k.assemble_residual(U, U1, F, cells=(0, min(k.C, 4096)))            # page in the tables / code
    t0, reps = time.perf_counter(), 0
    while reps < 1 or (time.perf_counter() - t0 < budget_s / 3 and reps < 5):
        F[:] = 0.0
        k.assemble_residual(U, U1, F)
        reps += 1
    dt_res = (time.perf_counter() - t0) / reps
    out["residual_assembly"] = {"seconds_per_assembly": dt_res, "tets_per_s": k.C / dt_res, "assemblies": reps,
                                "GBps_algorithmic": k.C * 1676.0 / dt_res / 1e9, "checksum": float(np.abs(F).sum())}
    # Jacobian element arithmetic: consecutive chunks from the middle of the mesh until a third of the budget is used
    chunk, c0 = max(cores * 16, 1024), k.C // 2
    t0, done = time.perf_counter(), 0
    while done == 0 or (time.perf_counter() - t0 < budget_s / 3 and c0 + chunk <= k.C):
        k.jacobian_elements(U, U1, c0, min(k.C, c0 + chunk))
        done += min(k.C, c0 + chunk) - c0
        c0 += chunk
    dt_jac = time.perf_counter() - t0
    out["jacobian_elements"] = {"tets": int(done), "seconds": dt_jac, "tets_per_s": done / dt_jac,
                                "seconds_for_the_whole_mesh_at_this_rate": k.C / (done / dt_jac),
                                "note": "complex-step element Jacobians of both parts of the split (64 complex residual evaluations "
                                        "per cell), no scatter into the CSR matrix"}
    # monolithic product on the mesh's own pattern
    est_nnz = 1530.0 * k.C
    if est_nnz <= max_nnz:
        t0 = time.perf_counter()
        indptr, indices, data = k.pattern()
        t_pat = time.perf_counter() - t0
        x, y = np.ones(k.ndof), np.empty(k.ndof)
        k.spmv(indptr, indices, data, x, y)
        t0, reps = time.perf_counter(), 0
        while reps < 2 or (time.perf_counter() - t0 < budget_s / 3 and reps < 20):
            k.spmv(indptr, indices, data, x, y)
            reps += 1
        dt_spmv = (time.perf_counter() - t0) / reps
        nnz = int(indptr[-1])
        out["spmv"] = {"nnz": nnz, "seconds_per_product": dt_spmv, "products": reps,
                       "GBps_algorithmic": (nnz * 12.0 + k.ndof * 20.0) / dt_spmv / 1e9, "pattern_build_s": t_pat,
                       "note": "CSR f64 + i32 on the bench mesh's own sparsity pattern (values: stand-ins, first touched in parallel)"}
    else:
        out["spmv"] = {"skipped": f"estimated {est_nnz:.2e} entries exceed the host-memory bound of this leg ({max_nnz:.1e})"}
    out["wall_s"] = time.perf_counter() - t_all
    return out
