"""Joint least-squares fit of the pin gap (the last bounded attempt).

    python tests/golden/make_pin_fit.py            # ~10 min on 8 cores -> tests/golden/pin_fit.npz

Every earlier scan (tests/test_pin_gap_study.py items 1-8) moved ONE knob.  Here all knobs move together: the response of the
sixteen pinned numbers - cylinder: v_x and d_x at vertex 0 for t = 1, 2, 3 ms and the predeformed vertex-0 coordinates (REF
tests/test_create_hdf5_and_separate_viz.py:40-51,196-206, tests/test_predeform.py:32-33); offset stenosis after five steps:
probe-5 velocity (3), pressure, solid-probe-5 displacement (3) (REF tests/test_simulations.py:34-35,53) - to

  physical parameters   rho_f, mu_f, rho_s, mu_s, lambda_s, interface-load scale, inlet scale, delta
  scheme parameters     theta, dt in the solid's d-v row only
  time-level weights    interface load lagged towards t^{n-1} (a = 1 - theta is the theta-weighted load), Laplace lifting term
                        on theta d^n + (1 - theta) d^{n-1} instead of d^n, fluid pressure theta-weighted instead of implicit

is linear to 1e-7 for small changes (checked below on one column), so the columns are finite differences of converged runs of
the numpy / C oracle.  Observations are in units of the REFERENCE'S OWN TOLERANCE, (value - pin) / (atol + 1e-5 |pin|): an entry
within [-1, 1] passes the reference's assert.  The file holds the baseline observations, the columns and the parameter names;
tests/test_pin_gap_study.py::test_joint_fit_of_all_parameters_and_time_level_weights does the fit and asserts its outcome.
"""
import contextlib
import copy
import io
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse.linalg as spla

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle.fsi_oracle as fo  # noqa: E402
from oracle.fsi_oracle import I3, FsiOracle, _inv3  # noqa: E402

CYL_V = np.array([4.38261949610407e-6, 5.244315455211961e-6, 8.137814761280497e-6])
CYL_D = np.array([2.235075700301419e-9, 7.0569699656660426e-9, 1.3776599148439903e-8])
CYL_PRE = np.array([7.382372340085156e-5, -1.1083576098054155e-4, 4.930899508039441e-4])
STE_V = np.array([-0.012555684636129378, 8.084632937234429e-06, -2.3712435710623827e-05])
STE_P = 0.43014573081840823
STE_D = np.array([-9.431090796213597e-06, -4.33478380630615e-05, -4.655061542874265e-05])
PARAMS = ["rho_f", "mu_f", "rho_s", "mu_s", "lambda_s", "load", "inlet", "delta", "theta", "k_dv", "load_lag", "laplace_lag", "pressure_lag"]
EPS = {"load_lag": 1e-2, "laplace_lag": 1e-2, "pressure_lag": 1e-2}          # absolute weights; everything else: relative 1e-3


class Variant(FsiOracle):
    """The oracle with three extra knobs (zero = the restated equations): ``k_dv`` relative change of dt in the solid's d-v
    row, ``laplace_lag`` b: grad d^n -> grad (d^n + b (d^{n-1} - d^n)) in the lifting term, ``pressure_lag`` c: p^n -> p^n +
    c (p^{n-1} - p^n) in the fluid's pressure stress."""
    k_dv = laplace_lag = pressure_lag = 0.0

    def _fluid_residual(self, cells, rho, mu, loc, loc1):
        Rl, Rn = super()._fluid_residual(cells, rho, mu, loc, loc1)
        if self.laplace_lag == 0.0 and self.pressure_lag == 0.0:
            return Rl, Rn
        d, v, p = self.unpack(loc)
        d1, v1, p1 = self.unpack(loc1)
        G, L, w = self.G[cells], self.L, self.wdet[cells]
        gd = np.einsum("cai,cqaj->cqij", d, G)
        gd1 = np.einsum("cai,cqaj->cqij", d1, G)
        zd, zp = np.zeros_like(d), np.zeros_like(p)
        if self.laplace_lag:
            Rl = Rl + self.pack(np.einsum("cq,cqaj,cqij->cai", w, G, self.laplace_lag * (gd1 - gd)), zd, zp)
        if self.pressure_lag:
            Finv, J = _inv3(I3 + gd)
            dp = self.pressure_lag * np.einsum("qa,ca->cq", L, p1 - p)
            grd = J[..., None, None] * (-dp[..., None, None] * np.swapaxes(Finv, -1, -2))
            Rn = Rn + self.pack(zd, np.einsum("cq,cqaj,cqij->cai", w, G, grd), zp)
        return Rl, Rn

    def _solid_residual(self, cells, rho, mu, lam, loc, loc1, **kw):
        Rl, Rn = super()._solid_residual(cells, rho, mu, lam, loc, loc1, **kw)
        if self.k_dv:
            d, v, p = self.unpack(loc)
            d1, _, _ = self.unpack(loc1)
            N, w = self.N, self.wdet[cells]
            dq = np.einsum("qa,cai->cqi", N, d - d1)
            extra = fo.DELTA * rho * (1.0 / (self.dt * (1.0 + self.k_dv)) - 1.0 / self.dt) * dq
            Rl = Rl + self.pack(np.einsum("cq,qa,cqi->cai", w, N, extra), np.zeros_like(d), np.zeros_like(p))
        return Rl, Rn


def make_oracle(desc, name, eps):
    """The oracle with parameter ``name`` moved by eps (relative, or absolute for the weights); returns (oracle, pscale, vscale, lag)."""
    d2 = copy.deepcopy(desc)
    pscale = vscale = 1.0
    lag = 0.0
    knobs = {}
    idx = {"rho_f": ("fluid_props", 0), "mu_f": ("fluid_props", 1), "rho_s": ("solid_props", 0), "mu_s": ("solid_props", 1),
           "lambda_s": ("solid_props", 2)}
    if name in idx:
        key, i = idx[name]
        rows = [list(p) for p in d2[key]]
        for p in rows:
            p[i] *= 1 + eps
        d2[key] = [tuple(p) for p in rows]
    elif name == "load":
        pscale = 1 + eps
    elif name == "inlet":
        vscale = 1 + eps
    elif name == "theta":
        d2["theta"] = float(d2["theta"]) * (1 + eps)
    elif name == "load_lag":
        lag = eps
    elif name in ("k_dv", "laplace_lag", "pressure_lag"):
        knobs[name] = eps
    use_numpy = bool(knobs)
    o = Variant(d2, impl="numpy") if use_numpy else FsiOracle(d2)
    for k, v in knobs.items():
        setattr(o, k, v)
    return o, pscale, vscale, lag


def converged_steps(o, lus, data, nsteps, pscale=1.0, vscale=1.0, lag=0.0, tol=1e-11, max_it=40, refactor=False, record=None):
    """``nsteps`` time steps from rest, each Newton-converged with the step's kept factorisation ``lus[k]`` (refactor: build them
    at the converged states of THIS run).  Returns the states after every step."""
    U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
    states, P_prev = [], 0.0
    for k in range(nsteps):
        g, P = data(k)
        g, Pe = g * vscale, (P + lag * (P_prev - P)) * pscale
        P_prev = P
        if refactor:                                  # Jacobian at the step's start state, refreshed once after two iterations
            lus[k] = spla.splu(o.jacobian(U, U1).tocsc())
        for it in range(max_it):
            dU = lus[k].solve(o.rhs(U, U1, Pe, g))
            U += dU
            U[o.bc_dofs] = g
            if refactor and it == 1:
                lus[k] = spla.splu(o.jacobian(U, U1).tocsc())
            if np.linalg.norm(dU) <= tol * max(np.linalg.norm(U), 1e-300):
                break
        else:
            raise RuntimeError(f"step {k} did not converge")
        states.append(U.copy())
        U1[:] = U
    return states


def tol_units(value, pin, atol):
    return (np.asarray(value) - np.asarray(pin)) / (atol + 1e-5 * np.abs(pin))


def main():
    from conftest import GOLDEN, prepare_case
    import tempfile
    from vasp_amd.fem import MixedFunction
    fo_delta = fo.DELTA
    out = {}
    t00 = time.time()
    for case, nsteps, dt in (("cylinder", 3, 1e-3), ("stenosis", 5, 1e-2)):
        if case == "cylinder":
            ns, desc, bc_values, pressure, hook = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tempfile.mkdtemp())
        else:
            ns, desc, bc_values, pressure, hook = prepare_case("offset_stenosis", GOLDEN / "offset_stenosis" / "offset_stenosis.h5",
                                                               tempfile.mkdtemp(), dt="0.01", T="0.04")
        mesh = ns["mesh"]

        def data(k, ns=ns, hook=hook, bc_values=bc_values, pressure=pressure, dt=dt):
            with contextlib.redirect_stdout(io.StringIO()):
                ns["t"] = dt * (k + 1)
                hook("pre_solve")(**ns)
            return bc_values(), float(pressure.P)

        def obs(states, o, case=case, mesh=mesh, ns=ns):
            if case == "cylinder":
                N2 = o.N2
                v = np.array([s[3 * N2] for s in states])
                d = np.array([s[0] for s in states])
                X0 = mesh.coords[0]
                return np.concatenate([tol_units(v, CYL_V, 1e-10), tol_units(d, CYL_D, 1e-10), tol_units(X0 - states[2][:3], CYL_PRE, 1e-10)])
            U = states[4]
            f = lambda pt, fld: MixedFunction(mesh, U).sub(fld)(pt)
            v, p, d = f(ns["probe_points"][5], 1), f(ns["probe_points"][5], 2), f(ns["solid_probe_points"][5], 0)
            return np.concatenate([tol_units(v, STE_V, 1e-8), [tol_units(p, STE_P, 1e-8)], tol_units(d, STE_D, 1e-8)])

        o0 = FsiOracle(desc)
        Z = np.zeros(o0.ndof)
        o0.solver_setup(Z, Z)
        lus = [None] * nsteps
        t0 = time.time()
        base = converged_steps(o0, lus, data, nsteps, refactor=True)
        y0 = obs(base, o0)
        print(f"{case}: baseline in {time.time() - t0:.0f} s; observations in tolerance units: {np.round(y0, 2)}", flush=True)
        cols = []
        for name in PARAMS:
            eps = EPS.get(name, 1e-3)
            t0 = time.time()
            if name == "delta":
                fo.DELTA = fo_delta * (1 + eps)
            try:
                o, ps, vs, lag = make_oracle(desc, name, eps)
                o.solver_setup(Z, Z)
                st = converged_steps(o, lus, data, nsteps, pscale=ps, vscale=vs, lag=lag)
            finally:
                fo.DELTA = fo_delta
            cols.append((obs(st, o) - y0) / eps)
            print(f"  {name:13s} {time.time() - t0:5.0f} s  column norm {np.linalg.norm(cols[-1]):.3e}", flush=True)
        # linearity check on one column: half the step gives the same derivative to 1e-3 of itself
        o, ps, vs, lag = make_oracle(desc, "mu_s", 5e-4)
        o.solver_setup(Z, Z)
        half = (obs(converged_steps(o, lus, data, nsteps), o) - y0) / 5e-4
        lin = np.linalg.norm(half - cols[PARAMS.index("mu_s")]) / np.linalg.norm(half)
        print(f"  linearity (mu_s, eps 1e-3 vs 5e-4): {lin:.2e}", flush=True)
        out[f"{case}_y0"], out[f"{case}_cols"], out[f"{case}_linearity"] = y0, np.array(cols).T, lin
    np.savez(ROOT / "tests" / "golden" / "pin_fit.npz", params=np.array(PARAMS), **out)
    print(f"done in {time.time() - t00:.0f} s")


if __name__ == "__main__":
    main()
