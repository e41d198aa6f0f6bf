"""GPU parity: the HIP path (through the C-ABI, vasp_amd.capi.HipBackend) against the CPU oracle, the committed golden
runs and the reference's known answers.  FP64 throughout; tolerances are written at each assert."""
import contextlib
import io

import numpy as np
import pytest

from conftest import GOLDEN, prepare_case

pytestmark = pytest.mark.gpu

# production defaults (mixed storage, forcing 1e-2) against exact solves on the known-answer case, per field; see
# test_production_storage_precisions_do_not_change_the_result (values measured on MI355X are in DESIGN.md section 2)
# measured (MI355X, round 3): 9.5e-7 / 5.8e-6 / 1.4e-6 and, with another preconditioner configuration, 3.3e-6 / 1.7e-5 / 5.6e-6:
# the distance is the policy's own stopping tolerance (update norm 1e-6, forcing 1e-2) accumulated over five steps, and it moves
# with whatever changes the inexact solves' error directions; bound = 3x the larger observation
# round 4: the last Newton iteration of a step is solved with the tighter forcing term 3e-3: measured
# 2.8e-7 / 1.4e-6 / 3.2e-7 (profiles/r04_forcing_scan.txt); the bound on v is now the reference's own rtol 1e-5
PRODUCTION_VS_EXACT_BOUND = {"d": 3e-6, "v": 1e-5, "p": 3e-6}


def random_state(mesh, ndof, seed=0):
    rng = np.random.default_rng(seed)
    N2, h = mesh.num_nodes, mesh.hmin()
    U, U1 = np.zeros(ndof), np.zeros(ndof)
    U[:3 * N2] = 0.02 * h * rng.standard_normal(3 * N2)
    U1[:3 * N2] = U[:3 * N2] + 0.002 * h * rng.standard_normal(3 * N2)
    U[3 * N2:6 * N2] = 0.1 * rng.standard_normal(3 * N2)
    U1[3 * N2:6 * N2] = U[3 * N2:6 * N2] + 0.01 * rng.standard_normal(3 * N2)
    U[6 * N2:] = 10 * rng.standard_normal(mesh.num_vertices)
    return U, U1


def boundary_data(case, t):
    ns, desc, bc_values, pressure, hook = case
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = t
        hook("pre_solve")(**ns)
    return bc_values(), (float(pressure.P) if pressure is not None else 0.0)


@pytest.fixture(scope="module")
def cyl(cylinder_case):
    from vasp_amd.capi import HipBackend
    hb = HipBackend(cylinder_case[1])
    yield hb
    hb.close()


@pytest.mark.parametrize("which", ["cylinder", "stenosis"])
def test_residual_matches_oracle(which, cylinder_case, stenosis_case):
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    case = cylinder_case if which == "cylinder" else stenosis_case
    desc, mesh = case[1], case[0]["mesh"]
    o = FsiOracle(desc)
    hb = HipBackend(desc)
    U, U1 = random_state(mesh, o.ndof)
    g, P = boundary_data(case, 0.05)
    hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    nrm = hb.assemble_residual()
    b_ref = o.rhs(U, U1, P, g)
    b = hb.get_state("b")
    assert np.abs(b - b_ref).max() <= 1e-12 * np.abs(b_ref).max()          # round-off of a different summation order
    assert np.isclose(nrm, np.linalg.norm(b_ref), rtol=1e-12)
    assert np.array_equal(hb.get_state("n"), U)                             # user <-> solver permutation is lossless
    hb.close()


def test_jacobian_spmv_and_solve_match_oracle(cyl, cylinder_case):
    import scipy.sparse.linalg as spla
    from oracle.fsi_oracle import FsiOracle
    desc, mesh = cylinder_case[1], cylinder_case[0]["mesh"]
    o = FsiOracle(desc)
    U, U1 = random_state(mesh, o.ndof, seed=1)
    g, P = boundary_data(cylinder_case, 0.05)
    cyl.set_state("n", U); cyl.set_state("n-1", U1); cyl.set_dirichlet_values(g); cyl.set_interface_pressure(P)
    cyl.assemble_residual()
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref = o.jacobian(U, U1)                                                # complex-step tangent, ident_zeros, bc rows
    cyl.assemble_jacobian()
    A = cyl.matrix()
    rowmax = np.maximum(np.abs(A_ref).max(axis=1).toarray().ravel(), 1e-300)
    rel = np.abs(A - A_ref).max(axis=1).toarray().ravel() / rowmax
    assert rel.max() < 1e-11                                                # forward-mode vs complex-step, entries to 1e13
    x = np.random.default_rng(2).standard_normal(o.ndof)
    y = cyl.spmv(x)
    assert np.abs(y - A_ref @ x).max() <= 1e-13 * np.abs(A_ref @ x).max() * 100
    # zero rows -> identity (pressure dofs that touch only solid cells), Dirichlet rows -> identity
    d = A.diagonal()
    assert np.all(d[o.zero_rows] == 1.0) and np.all(d[o.bc_dofs] == 1.0)
    # linear solve at a physical state (the committed converged run): Newton update vs sparse LU of the oracle's matrix
    gold = np.load(GOLDEN / "cylinder_tight.npz")["states"]
    U, U1 = gold[1].copy(), gold[0].copy()
    g, P = boundary_data(cylinder_case, 3e-3)                                # data of the next step: a non-trivial rhs
    cyl.set_state("n", U); cyl.set_state("n-1", U1); cyl.set_dirichlet_values(g); cyl.set_interface_pressure(P)
    cyl.assemble_residual()
    cyl.assemble_jacobian()
    A_ref = o.jacobian(U, U1)
    b_ref = o.rhs(U, U1, P, g)
    it, rr = cyl.solve(lin_rtol=1e-11)
    du_ref = spla.splu(A_ref.tocsc()).solve(b_ref)
    du = cyl.get_state("du")
    N2 = mesh.num_nodes
    for sl in (slice(0, 3 * N2), slice(3 * N2, 6 * N2), slice(6 * N2, None)):
        assert np.linalg.norm(du[sl] - du_ref[sl]) <= 1e-5 * np.linalg.norm(du_ref[sl])   # cond(A) ~ 1e12: LU itself carries ~1e-5
    cyl.set_state("n", np.zeros(o.ndof)); cyl.set_state("n-1", np.zeros(o.ndof))


@pytest.mark.parametrize("lin_solver,precond", [
    pytest.param(1, 1, marks=pytest.mark.xfail(strict=True, reason="BiCGStab + monolithic ILU(0) does not converge on the FSI Jacobian: "
                                                                     "relres 1.4e-1 after 3 000, 8.9e-4 after 20 000 iterations on the "
                                                                     "1 647-tet cylinder (MI355X, round 4); each option converges with the "
                                                                     "other's default partner - DESIGN.md section 5")),
    (1, 0), (0, 1)])
def test_bicgstab_and_ilu0_solve_match_sparse_lu(cylinder_case, monkeypatch, lin_solver, precond):
    """The solver `north_star` names literally - BiCGStab with a (multicolour) ILU(0) preconditioner - is reachable through
    the C-ABI (FsiNewtonOpts.lin_solver = 1, fsi_set_linear_solver precond = 1; include/vaspfsi.h) and is held to the same
    check as the default GCR / field-split pair: the Newton update of a physical state against a sparse LU of the oracle's
    matrix.  (1, 1) is that solver: it stagnates on this matrix (a saddle point with the 1e7 penalty rows: the pressure
    block has no diagonal for ILU(0) to work with), which the strict xfail records; (1, 0) BiCGStab with the field-split
    preconditioner and (0, 1) recycled GCR with ILU(0) both converge to 1e-10 and reproduce the sparse-LU update and the
    golden time step.  VERDICT r3 item 7: no untested solver option behind include/vaspfsi.h."""
    import scipy.sparse.linalg as spla
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    monkeypatch.setenv("FSI_ORDER", "colour")            # the ILU(0) kernels need the multicolour node ordering
    desc, mesh = cylinder_case[1], cylinder_case[0]["mesh"]
    o = FsiOracle(desc)
    hb = HipBackend(desc, lin_solver=lin_solver, precond=precond)
    gold = np.load(GOLDEN / "cylinder_tight.npz")["states"]
    U, U1 = gold[1].copy(), gold[0].copy()
    g, P = boundary_data(cylinder_case, 3e-3)
    hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    hb.assemble_jacobian()
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref, b_ref = o.jacobian(U, U1), o.rhs(U, U1, P, g)
    from vasp_amd.capi import FsiError
    try:
        it, rr = hb.solve(lin_rtol=1e-10, lin_max_it=3000 if (lin_solver, precond) == (1, 1) else 20000)
    except FsiError as e:
        hb.close()
        pytest.fail(f"lin_solver {lin_solver} precond {precond}: {e}")
    print(f"lin_solver {lin_solver} precond {precond}: {it} iterations, relres {rr:.2e}")
    assert rr <= 1e-10
    du_ref = spla.splu(A_ref.tocsc()).solve(b_ref)
    du = hb.get_state("du")
    N2 = mesh.num_nodes
    for sl in (slice(0, 3 * N2), slice(3 * N2, 6 * N2), slice(6 * N2, None)):
        assert np.linalg.norm(du[sl] - du_ref[sl]) <= 1e-5 * np.linalg.norm(du_ref[sl])
    # and a whole time step through the Newton driver with that solver: same state as the default solver's converged golden run
    hb.set_state("n", gold[1]); hb.set_state("n-1", gold[1])
    hb.lin_rtol = 1e-11
    hist = hb.newton_solve(counter=2, first_step_num=0, atol=1e-11, rtol=1e-14, max_it=30, lmbda=1.0, recompute=20, recompute_tstep=20)
    assert hist[-1][0] < 1e-11 or hist[-1][1] < 1e-14
    Un = hb.get_state("n")
    for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
        err = np.linalg.norm(Un[sl] - gold[2][sl]) / np.linalg.norm(gold[2][sl])
        assert err < 1e-6, (name, err)
    hb.close()


def test_cylinder_three_steps_match_converged_golden(cylinder_case):
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = cylinder_case
    mesh = ns["mesh"]
    gold = np.load(GOLDEN / "cylinder_tight.npz")["states"]
    hb = HipBackend(desc, lin_rtol=1e-11)
    N2 = mesh.num_nodes
    for k in range(3):
        g, P = boundary_data(cylinder_case, 1e-3 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=1e-11, rtol=1e-14, max_it=30, lmbda=1.0, recompute=20,
                               recompute_tstep=20)
        assert hist[-1][0] < 1e-11 or hist[-1][1] < 1e-14
        hb.shift()
        U = hb.get_state("n")
        for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
            err = np.linalg.norm(U[sl] - gold[k][sl]) / np.linalg.norm(gold[k][sl])
            assert err < 1e-7, (k, name, err)                                # both sides solve F = 0 to 1e-11
    hb.close()


@pytest.fixture(scope="module")
def known_answer_run(stenosis_case):
    """The reference's primary known-answer case [REF tests/test_simulations.py:17-57] through the HIP path with the
    reference's Newton policy: tolerances 1e-6, Jacobian reuse, and every Newton system solved (forcing = 0: to 1e-10, the
    stand-in for the reference's direct LU).  Returns (final state, Newton iterations per step)."""
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = stenosis_case
    hb = HipBackend(desc, lin_rtol=1e-10, newton_forcing=0.0)
    its = []
    for k in range(5):
        g, P = boundary_data(stenosis_case, 0.01 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        h = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                            recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
        its.append(len(h))
        hb.shift()
    U = hb.get_state("n")
    hb.close()
    return U, its


def test_offset_stenosis_known_answer_on_gpu(stenosis_case, known_answer_run):
    """The HIP path must follow the ORACLE's run of the reference's policy (tests/golden/stenosis_ref.npz) iteration by
    iteration; against the reference's pins it inherits the oracle's measured gap (tests/test_oracle_pins.py), bounded here
    at the same values.  The reference's own tolerances are held by the strict-xfail test below."""
    from test_oracle_pins import PIN_D, PIN_P, PIN_V, probe
    ns, mesh = stenosis_case[0], stenosis_case[0]["mesh"]
    U, its = known_answer_run
    gold = np.load(GOLDEN / "stenosis_ref.npz")
    assert its == [int(i) for i in gold["iterations"]], (its, gold["iterations"])      # same quasi-Newton trajectory: 3 4 8 11 5
    G = gold["states"][4]
    N2 = mesh.num_nodes
    for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
        err = np.linalg.norm(U[sl] - G[sl]) / np.linalg.norm(G[sl])
        assert err < 2e-6, (name, err)                        # the last Newton update of the policy is ~1e-6; both runs stop there
    v = probe(mesh, U, ns["probe_points"][5], 1)
    p = probe(mesh, U, ns["probe_points"][5], 2)
    d = probe(mesh, U, ns["solid_probe_points"][5], 0)
    assert np.abs(v - PIN_V).max() < 4.8e-7, (v, PIN_V)
    assert abs(p - PIN_P) < 3.9e-4, (p, PIN_P)
    assert np.abs(d - PIN_D).max() < 1.4e-8, (d, PIN_D)


@pytest.mark.xfail(strict=True, reason="the restated equations miss the reference's pins by 2.6e-5 (v) / 7e-4 (p); "
                                       "DESIGN.md section 2 - flips together with tests/test_oracle_pins.py")
def test_reference_tolerance_on_offset_stenosis_pins_on_gpu(stenosis_case, known_answer_run):
    """The reference's asserts VERBATIM [REF tests/test_simulations.py:43-44,57: np.isclose defaults, rtol 1e-5 / atol 1e-8]
    on the HIP path's output.  VERDICT r2 item 1a: the HIP side must flip with the oracle the day the gap closes."""
    from test_oracle_pins import PIN_D, PIN_P, PIN_V, probe
    ns, mesh = stenosis_case[0], stenosis_case[0]["mesh"]
    U, _ = known_answer_run
    v = probe(mesh, U, ns["probe_points"][5], 1)
    p = probe(mesh, U, ns["probe_points"][5], 2)
    d = probe(mesh, U, ns["solid_probe_points"][5], 0)
    assert np.isclose(v, PIN_V).all()
    assert np.isclose(p, PIN_P)
    assert np.isclose(d, PIN_D).all()


@pytest.fixture(scope="module")
def production_run(stenosis_case):
    """The same five known-answer steps with what SHIPS: HipBackend's defaults (inexact Newton with the default forcing
    terms, FP32 Krylov basis and Jacobian copy inside the iterations, FP16 / FP32 preconditioner matrices) - VERDICT r3 item
    1a: the forcing-0 run above is the stand-in for the reference's direct LU, this one is the product."""
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = stenosis_case
    hb = HipBackend(desc)
    its, kry = [], 0
    for k in range(5):
        g, P = boundary_data(stenosis_case, 0.01 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        h = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                            recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
        its.append(len(h)); kry += sum(it[3] for it in h)
        hb.shift()
    U = hb.get_state("n")
    tm = hb.timers()
    hb.close()
    return U, its, kry, tm


def test_offset_stenosis_known_answer_with_production_defaults(stenosis_case, production_run):
    """The shipped defaults held to the reference's pins at the bounds the oracle's own run of the policy is held to
    (tests/test_oracle_pins.py: 4.8e-7 / 3.9e-4 / 1.4e-8), with the oracle's Newton trajectory 3 4 8 11 5."""
    from test_oracle_pins import PIN_D, PIN_P, PIN_V, probe
    ns, mesh = stenosis_case[0], stenosis_case[0]["mesh"]
    U, its, kry, tm = production_run
    assert tm["q_elem_bytes"] == 4 and tm["spmv_fp32_calls"] > 0            # the defaults did select the mixed storage
    gold = np.load(GOLDEN / "stenosis_ref.npz")
    assert its == [int(i) for i in gold["iterations"]], (its, gold["iterations"])
    v = probe(mesh, U, ns["probe_points"][5], 1)
    p = probe(mesh, U, ns["probe_points"][5], 2)
    d = probe(mesh, U, ns["solid_probe_points"][5], 0)
    print("production defaults vs pins: |v - pin| %.3e  |p - pin| %.3e  |d - pin| %.3e  (Krylov iterations %d, late solves %d)"
          % (np.abs(v - PIN_V).max(), abs(p - PIN_P), np.abs(d - PIN_D).max(), kry, tm["newton_late_solves"]))
    assert np.abs(v - PIN_V).max() < 4.8e-7, (v, PIN_V)
    assert abs(p - PIN_P) < 3.9e-4, (p, PIN_P)
    assert np.abs(d - PIN_D).max() < 1.4e-8, (d, PIN_D)


@pytest.mark.xfail(strict=True, reason="the restated equations miss the reference's pins by 2.6e-5 (v) / 7e-4 (p) whatever solves them; "
                                       "DESIGN.md section 2 - flips together with tests/test_oracle_pins.py")
def test_reference_tolerance_on_offset_stenosis_pins_with_production_defaults(stenosis_case, production_run):
    """The reference's asserts VERBATIM [REF tests/test_simulations.py:43-44,57] on the output of the shipped defaults."""
    from test_oracle_pins import PIN_D, PIN_P, PIN_V, probe
    ns, mesh = stenosis_case[0], stenosis_case[0]["mesh"]
    U = production_run[0]
    v = probe(mesh, U, ns["probe_points"][5], 1)
    p = probe(mesh, U, ns["probe_points"][5], 2)
    d = probe(mesh, U, ns["solid_probe_points"][5], 0)
    assert np.isclose(v, PIN_V).all()
    assert np.isclose(p, PIN_P)
    assert np.isclose(d, PIN_D).all()


def test_forced_fp32_basis_reports_instead_of_overrunning(cylinder_case, monkeypatch):
    """FSI_KRYLOV_FP32=1 sizes the basis store for 4-byte columns.  A tolerance FP32 storage cannot reach used to
    switch to 8-byte columns in that same store (writes past the allocation); now the solve either succeeds in FP32 or
    returns FSI_ERR_LINEAR with a message that names the switch - and the context stays usable."""
    from vasp_amd.capi import FsiError, HipBackend
    monkeypatch.setenv("FSI_KRYLOV_FP32", "1")
    hb = HipBackend(cylinder_case[1], lin_rtol=1e-14, newton_forcing=0.0)
    g, P = boundary_data(cylinder_case, 1e-3)
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual(); hb.assemble_jacobian()
    assert hb.timers()["q_elem_bytes"] == 4
    try:
        it, rr = hb.solve(lin_rtol=1e-14, lin_max_it=400)
        assert rr <= 1e-12                                   # it may get there through restarts; then the answer is good
    except FsiError as e:
        assert e.code == 4 and ("FSI_KRYLOV_FP32" in str(e) or "GCR" in str(e)), (e.code, str(e))
    it, rr = hb.solve(lin_rtol=1e-6)                         # the context is intact: a tolerance FP32 storage reaches
    assert rr <= 1e-6
    assert hb.timers()["q_elem_bytes"] == 4                  # and it never left the 4-byte layout
    hb.close()


def test_production_storage_precisions_do_not_change_the_result(stenosis_case, monkeypatch):
    """The production defaults (FP32 Krylov basis and FP32 copy of the Jacobian inside the iterations, FP16 / FP32 matrix
    copies in the preconditioner sweeps, inexact-Newton forcing 1e-2) against the same five steps with every one of them
    switched off (all FP64, every system to 1e-10): the reference's Newton policy (tolerances 1e-6) must stop on the same
    fields - the storage precisions only exist inside linear solves whose answers are judged on the FP64 matrix."""
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = stenosis_case
    mesh = ns["mesh"]

    def run(**kw):
        hb = HipBackend(desc, **kw)
        its, kry = [], 0
        for k in range(5):
            g, P = boundary_data(stenosis_case, 0.01 * (k + 1))
            hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
            h = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                                recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
            its.append(len(h)); kry += sum(it[3] for it in h)
            hb.shift()
        U = hb.get_state("n")
        tm = hb.timers()
        hb.close()
        return U, its, kry, tm

    U_prod, its_prod, kry_prod, tm = run()                                  # defaults
    assert tm["q_elem_bytes"] == 4 and tm["spmv_fp32_calls"] > 0            # ... which did select the FP32 basis and operator
    for var in ("FSI_KRYLOV_FP32", "FSI_OPERATOR_FP32", "FSI_SCHUR_FP32", "FSI_SWEEPS_FP16"):
        monkeypatch.setenv(var, "0")
    U_ref, its_ref, kry_ref, tm_ref = run(lin_rtol=1e-10, newton_forcing=0.0)
    assert tm_ref["q_elem_bytes"] == 8 and tm_ref["spmv_fp32_calls"] == 0
    N2 = mesh.num_nodes
    errs = {}
    for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
        errs[name] = float(np.linalg.norm(U_prod[sl] - U_ref[sl]) / np.linalg.norm(U_ref[sl]))
    print("production defaults vs exact solves, relative l2 distance per field:", errs)
    print("Newton iterations per step", its_prod, "vs exact solves", its_ref, "| Krylov iterations", kry_prod, "vs", kry_ref)
    with contextlib.suppress(OSError):                 # kept with the round's measurements
        import json
        from conftest import ROOT
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / "production_vs_exact.json").write_text(json.dumps(
            dict(errors=errs, newton_prod=its_prod, newton_exact=its_ref, krylov_prod=kry_prod, krylov_exact=kry_ref)))
    # Bounds = the measured distances (MI355X, round 3) with 2x head room; they are the policy's own stopping tolerance
    # (1e-6 on the update, inexact solves with forcing 1e-2) accumulated over five steps, per field:
    for name, bound in PRODUCTION_VS_EXACT_BOUND.items():
        assert errs[name] < bound, (name, errs[name], bound)


def test_robin_terms_match_oracle(tmp_path):
    """aneurysm-style Robin boundary term on the outer wall [REF src/vasp/simulations/aneurysm.py:73-76]."""
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    case = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tmp_path)
    ns, desc = case[0], dict(case[1])
    mesh = ns["mesh"]
    fids = np.nonzero(ns["boundaries"] == 33)[0]
    desc["robin_facets"] = mesh.facet_nodes[fids]
    desc["robin_k"] = np.full(len(fids), 1e5)
    desc["robin_c"] = np.full(len(fids), 10.0)
    o = FsiOracle(desc)
    hb = HipBackend(desc)
    U, U1 = random_state(mesh, o.ndof, seed=3)
    g, P = boundary_data(case, 0.05)
    hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    b_ref = o.rhs(U, U1, P, g)
    assert np.abs(hb.get_state("b") - b_ref).max() <= 1e-12 * np.abs(b_ref).max()
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref = o.jacobian(U, U1)
    hb.assemble_jacobian()
    x = np.random.default_rng(4).standard_normal(o.ndof)
    assert np.abs(hb.spmv(x) - A_ref @ x).max() <= 1e-11 * np.abs(A_ref @ x).max()
    hb.close()


def test_properties_on_generated_mesh(tmp_path):
    """Size-independent properties on a synthetic ~50 k-tet offset stenosis (no oracle run at this size)."""
    from vasp_amd.capi import HipBackend
    from vasp_amd.meshgen import write_mesh
    write_mesh(tmp_path / "s.h5", 50000)
    case = prepare_case("offset_stenosis", tmp_path / "s.h5", tmp_path / "run", dt="0.001", T="0.002")
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    hb = HipBackend(desc)
    # F(0; 0) = 0 with homogeneous data
    hb.set_dirichlet_values(np.zeros(len(desc["bc_dofs"]))); hb.set_interface_pressure(0.0)
    assert hb.assemble_residual() == 0.0
    # SpMV is linear and reproducible
    g, P = boundary_data(case, 1e-3)
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_jacobian()
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(hb.ndof), rng.standard_normal(hb.ndof)
    ax, ay, axy = hb.spmv(x), hb.spmv(y), hb.spmv(2 * x - 3 * y)
    assert np.abs(axy - (2 * ax - 3 * ay)).max() <= 1e-12 * np.abs(axy).max()
    assert np.array_equal(hb.spmv(x), ax)
    # one time step: Newton residuals fall, and in the solid d = d1 + dt (theta v + (1 - theta) v1) holds nodally
    hist = hb.newton_solve(counter=0, first_step_num=0, atol=1e-9, rtol=1e-12, max_it=20, lmbda=1.0, recompute=20,
                           recompute_tstep=20)
    res = [h[0] for h in hist]
    assert res[-1] < 1e-3 * res[0]
    U = hb.get_state("n")
    d, v, _ = mesh.split(U)
    solid_only = np.setdiff1d(np.unique(mesh.tet_nodes[mesh.cell_markers == 2]), np.unique(mesh.tet_nodes[mesh.cell_markers != 2]))
    free = np.setdiff1d(solid_only, np.unique(desc["bc_dofs"][desc["bc_dofs"] < 3 * mesh.num_nodes] // 3))
    assert np.abs(d[free] - 1e-3 * ns["theta"] * v[free]).max() <= 1e-4 * np.abs(d[free]).max()   # Newton atol 1e-9
    hb.close()


@pytest.mark.parametrize("which", ["fixture", "generated"])
def test_exact_coarse_solve_matches_a_dense_factorisation(which, stenosis_case, tmp_path):
    """The solid cycle's coarse level solved by block cyclic reduction (csrc/fsi_bcr.hip; round 5) against a sparse LU of the
    same operator, read back from the device: offset-stenosis fixture (an unstructured wall) and a generated 50 k-tet mesh.
    Operators are FP32, vectors FP64: the residual of the answer sits at FP32 round-off times the level's condition number."""
    import scipy.sparse.linalg as spla
    from vasp_amd.capi import HipBackend
    if which == "fixture":
        case = stenosis_case
    else:
        from vasp_amd.meshgen import write_mesh
        write_mesh(tmp_path / "s.h5", 50000)
        case = prepare_case("offset_stenosis", tmp_path / "s.h5", tmp_path / "run", dt="0.001", T="0.002")
    desc = case[1]
    g, P = boundary_data(case, 0.01 if which == "fixture" else 1e-3)
    # what ships solves A_c + shift * blockdiag(A_c) (FsiTuning.bcr_shift: the lowest modes are damped, not inverted): held to
    # the sparse LU of THAT operator first, on one right-hand side
    import scipy.sparse as sp
    hb = HipBackend(desc)
    shift = hb.tuning()["bcr_shift"]
    assert 0.0 < shift < 1e-3
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    hb.assemble_jacobian()
    A, cptr, ccol = hb.solid_coarse_matrix()
    nc = A.shape[0] // 3
    diag = sp.block_diag([A[3 * i:3 * i + 3, 3 * i:3 * i + 3] for i in range(nc)], format="csr")
    rhs = np.random.default_rng(5).standard_normal(A.shape[0])
    x_ref = spla.splu((A + shift * diag).tocsc()).solve(rhs)
    x = hb.solid_coarse_solve(rhs)
    assert np.linalg.norm(x - x_ref) < 2e-2 * np.linalg.norm(x_ref)
    hb.close()
    # the rest with shift = 0: the level itself
    hb = HipBackend(desc, tuning=dict(bcr_shift=0.0))
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    hb.assemble_jacobian()
    info = hb.solid_coarse_info()
    assert info["cycle_ready"] == 1 and info["planned"] == 1 and info["ready"] == 1, info
    assert info["levels"] == int(np.ceil(np.log2(info["bfs_blocks"]))) and info["launches_per_solve"] == 2 * info["levels"] + 1
    A, cptr, ccol = hb.solid_coarse_matrix()
    n = A.shape[0]
    rng = np.random.default_rng(0)
    lu = spla.splu(A.tocsc())
    for trial in range(3):
        rhs = rng.standard_normal(n)
        if trial == 2:                                   # a smooth right-hand side: the low modes the level exists for
            rhs = A @ np.ones(n)
        x = hb.solid_coarse_solve(rhs)
        x_ref = lu.solve(rhs)
        res = np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs)
        err = np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref)
        assert res < 2e-4 and err < 2e-2, (which, trial, res, err)
    # linear and repeatable: the same operator at every call (what the outer Krylov method relies on)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    xa, xb, xab = hb.solid_coarse_solve(a), hb.solid_coarse_solve(b), hb.solid_coarse_solve(2 * a - 3 * b)
    assert np.abs(xab - (2 * xa - 3 * xb)).max() <= 1e-5 * np.abs(xab).max()
    assert np.array_equal(hb.solid_coarse_solve(a), xa)
    # and it is what an application of the preconditioner runs
    before = hb.solid_coarse_info()["solves"]
    hb.apply_preconditioner(rng.standard_normal(hb.ndof))
    assert hb.solid_coarse_info()["solves"] == before + 1
    hb.close()
    # FsiTuning.solid_coarse_exact = 0 keeps the sweeps
    hb = HipBackend(desc, tuning=dict(solid_coarse_exact=0))
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    hb.assemble_jacobian()
    assert hb.solid_coarse_info()["planned"] == 0
    hb.apply_preconditioner(rng.standard_normal(hb.ndof))
    assert hb.solid_coarse_info()["solves"] == 0
    hb.close()


def test_properties_at_bench_size(tmp_path):
    """BASELINE config 2 size (the 1.12 M-tet mesh bench.py times; no oracle run at this size): size-independent properties
    through the C-ABI - F(0; 0) = 0, linearity and reproducibility of the outer product, the FP32 working copy of the Jacobian
    against the FP64 one, and one time step of the production policy (FP32 basis / operator, FP16 sweep records) whose Newton
    residuals fall and whose solid obeys d = dt (theta v + (1 - theta) v1) nodally."""
    from vasp_amd.capi import HipBackend
    from vasp_amd.meshgen import write_mesh
    write_mesh(tmp_path / "s.h5", 1000000)
    case = prepare_case("offset_stenosis", tmp_path / "s.h5", tmp_path / "run", dt="0.001", T="0.002")
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    assert mesh.num_cells > 1_000_000
    hb = HipBackend(desc)
    hb.set_dirichlet_values(np.zeros(len(desc["bc_dofs"]))); hb.set_interface_pressure(0.0)
    assert hb.assemble_residual() == 0.0
    g, P = boundary_data(case, 1e-3)
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_jacobian()
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(hb.ndof), rng.standard_normal(hb.ndof)
    ax, ay, axy = hb.spmv(x), hb.spmv(y), hb.spmv(2 * x - 3 * y)
    assert np.abs(axy - (2 * ax - 3 * ay)).max() <= 1e-12 * np.abs(axy).max()
    assert np.array_equal(hb.spmv(x), ax)
    hist = hb.newton_solve(counter=0, first_step_num=0, atol=1e-6, rtol=1e-6, max_it=20, lmbda=1.0, recompute=20,
                           recompute_tstep=20)
    tm = hb.timers()
    assert tm["q_elem_bytes"] == 4 and tm["spmv_fp32_calls"] > 0           # the production storage precisions were in use
    res = [h[0] for h in hist]
    assert len(hist) <= 6 and (hist[-1][0] < 1e-6 or hist[-1][1] < 1e-6)    # the reference's stopping rule was met
    assert res[-1] < 1e-2 * res[0]
    U = hb.get_state("n")
    d, v, _ = mesh.split(U)
    solid_only = np.setdiff1d(np.unique(mesh.tet_nodes[mesh.cell_markers == 2]), np.unique(mesh.tet_nodes[mesh.cell_markers != 2]))
    free = np.setdiff1d(solid_only, np.unique(desc["bc_dofs"][desc["bc_dofs"] < 3 * mesh.num_nodes] // 3))
    assert np.abs(d[free] - 1e-3 * ns["theta"] * v[free]).max() <= 1e-3 * np.abs(d[free]).max()   # Newton stops at 1e-6
    hb.close()


def test_properties_at_config3_size_with_the_robin_wall(tmp_path):
    """BASELINE configs[2]: the aneurysm problem file (StVK wall, Robin condition on the outer wall [REF
    src/vasp/simulations/aneurysm.py:73-76], its own tolerances 1e-10 / 1e-9 [:48-49]) at ~3 M tets in ONE context - the size the
    config spreads over 4 GPUs - on the synthetic tube (the tutorial mesh is not in the tree).  No oracle run at this size:
    F(0; 0) = 0, linearity of the product, the Robin term's contribution to residual and matrix (a rigid wall translation d = c
    loads the outer wall with theta k_s c A; its derivative is the matrix column sum), and two time steps whose Newton residuals
    fall under the problem's tolerances with the storage the policy picks.  The 288 GB of one MI355X hold it: the HBM in use
    is printed."""
    import json
    from vasp_amd.capi import HipBackend
    from vasp_amd.mesh import FsiMesh
    from vasp_amd.meshgen import generate
    m = generate(3600000)                                # the generator's sizes: 2.60 M or 3.58 M tets; config 3 says ~3 M
    FsiMesh.from_arrays(m["coords"], m["tets"], m["cell_markers"], m["facets"], m["facet_markers"]).write(tmp_path / "aneurysm.h5")
    del m
    (tmp_path / "aneurysm_probe_point.json").write_text(json.dumps([[0.0, 0.0, 0.0], [16.0, 0.0, 0.0]]))
    case = prepare_case("aneurysm", tmp_path / "aneurysm.h5", tmp_path / "run", dt="0.001", T="0.2", theta="0.501")
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    assert mesh.num_cells > 3_000_000 and len(desc["robin_facets"]) > 0
    hb = HipBackend(desc)
    free_b, total_b = hb.device_memory()
    print(f"{mesh.num_cells} tets, {hb.ndof} dofs, matrix entries {int(hb.lib.fsi_matrix_nnz(hb.ctx))}: HBM in use {(total_b - free_b) / 2**30:.0f} GiB "
          f"of {total_b / 2**30:.0f}, Krylov capacity {hb.timers()['krylov_cap']}")
    nbc = len(desc["bc_dofs"])
    hb.set_dirichlet_values(np.zeros(nbc)); hb.set_interface_pressure(0.0)
    assert hb.assemble_residual() == 0.0
    # Robin term: with the solid displaced rigidly by c (fluid at rest, no velocity) the only load on the d-rows of the outer
    # wall... is carried by the v-equation rows: sum over the wall's v_x rows of F = theta k_s c_x A (partition of unity)
    N2 = mesh.num_nodes
    wall_nodes = np.unique(np.asarray(desc["robin_facets"]))
    solid_nodes = np.unique(mesh.tet_nodes[np.asarray(desc["cell_kind"]) == 1])
    U = np.zeros(hb.ndof)
    c = 1e-6
    U[3 * solid_nodes] = c                              # d_x = c on every solid node: zero strain, zero stress
    hb.set_state("n", U); hb.set_state("n-1", U)
    hb.assemble_residual()
    b = hb.get_state("b")
    x = mesh.coords
    f = np.asarray(desc["robin_facets"])[:, :3]
    area = 0.5 * np.linalg.norm(np.cross(x[f[:, 1]] - x[f[:, 0]], x[f[:, 2]] - x[f[:, 0]]), axis=1).sum()
    k_s = float(np.asarray(desc["robin_k"])[0])
    bc_set = set(np.asarray(desc["bc_dofs"]).tolist())
    rows = np.array([3 * N2 + 3 * n for n in wall_nodes if (3 * N2 + 3 * n) not in bc_set])
    # b = -F; the theta-weighted Robin load k_s (theta d^n + (1 - theta) d^{n-1}) = k_s c on the whole wall
    total = -b[3 * N2 + 3 * solid_nodes].sum()
    assert total == pytest.approx(k_s * c * area, rel=2e-2), (total, k_s * c * area)       # (Dirichlet rims of the wall are excluded)
    assert len(rows) > 0
    hb.set_state("n", np.zeros(hb.ndof)); hb.set_state("n-1", np.zeros(hb.ndof))
    g, P = boundary_data(case, 1e-3)
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_jacobian()
    rng = np.random.default_rng(0)
    xv, yv = rng.standard_normal(hb.ndof), rng.standard_normal(hb.ndof)
    ax, ay, axy = hb.spmv(xv), hb.spmv(yv), hb.spmv(2 * xv - 3 * yv)
    assert np.abs(axy - (2 * ax - 3 * ay)).max() <= 1e-12 * np.abs(axy).max()
    its = []
    for k in range(2):
        g, P = boundary_data(case, 1e-3 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                               recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
        hb.shift()
        assert hist[-1][0] < ns["atol"] or hist[-1][1] < ns["rtol"], hist
        assert hist[-1][0] < 1e-3 * hist[0][0]
        its.append([h[3] for h in hist])
    tm = hb.timers()
    print("Krylov iterations per Newton iteration:", its, "Q bytes", tm["q_elem_bytes"], "events",
          {k: int(tm[k]) for k in ("gcr_restarts", "newton_retries", "fp32_fallbacks")})
    assert tm["newton_retries"] == 0 and np.isfinite(hb.get_state("n")).all()
    hb.close()


def test_error_behaviour(cyl, cylinder_case):
    from vasp_amd.capi import FsiError
    with pytest.raises(FsiError) as e:
        cyl.set_dirichlet_values(np.zeros(3))                                # wrong length
    assert e.value.code == 1
    # a non-finite state makes the Newton loop report divergence, as the reference's RuntimeError
    U = np.zeros(cyl.ndof)
    U[5] = np.nan
    cyl.set_state("n", U)
    g, P = boundary_data(cylinder_case, 1e-3)
    cyl.set_dirichlet_values(g)
    with pytest.raises((RuntimeError, FsiError)):
        cyl.newton_solve(counter=0, first_step_num=0, atol=1e-6, rtol=1e-6, max_it=3, lmbda=1.0, recompute=20, recompute_tstep=20)
    cyl.set_state("n", np.zeros(cyl.ndof)); cyl.set_state("n-1", np.zeros(cyl.ndof))


def test_aneurysm_runs_and_prints_sane_flow_properties(tmp_path):
    """REF tests/test_simulations.py:80-125: the aneurysm problem (Robin wall) runs; velocity, CFL and Reynolds numbers
    printed by post_solve are finite and non-negative (the reference pins nothing else for this problem)."""
    import re
    from vasp_amd import monolithic
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        monolithic.run(["-p", "aneurysm", "-dt", "0.001", "-T", "0.002", "--theta", "0.51", "--folder", str(tmp_path), "--sub-folder", "1",
                        "--new-arguments", f"mesh_path={GOLDEN / 'aneurysm' / 'small_aneurysm.h5'}", "inlet_id=4"], out=print)
    out = buf.getvalue()
    for pat in (r"Velocity \(mean, min, max\): (.*), (.*), (.*)", r"CFL \(mean, min, max\): (.*), (.*), (.*)",
                r"Reynolds Numbers \(mean, min, max\): (.*), (.*), (.*)"):
        m = re.findall(pat, out)
        assert len(m) == 3
        vals = np.array([[float(x) for x in row] for row in m])
        assert np.all(np.isfinite(vals)) and np.all(vals >= 0)
    assert out.count("Solved for timestep") == 3


def test_mooney_rivlin_matches_oracle(tmp_path):
    """predeform problem (MooneyRivlin + Robin, theta = 1): residual and Jacobian of the HIP path vs the oracle."""
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    case = prepare_case("predeform", GOLDEN / "cylinder" / "cylinder.h5", tmp_path, dt="0.01", T="0.02", theta="1.0")
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    o = FsiOracle(desc)
    hb = HipBackend(desc)
    rng = np.random.default_rng(7)
    N2, h = mesh.num_nodes, mesh.hmin()
    U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
    U[:3 * N2] = 0.01 * h * rng.standard_normal(3 * N2)                  # a few per cent strain
    U1[:3 * N2] = 0.9 * U[:3 * N2]
    U[3 * N2:6 * N2] = 0.05 * rng.standard_normal(3 * N2)
    U[6 * N2:] = rng.standard_normal(mesh.num_vertices)
    g, P = boundary_data(case, 0.5)
    hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    b_ref = o.rhs(U, U1, P, g)
    assert np.abs(hb.get_state("b") - b_ref).max() <= 1e-11 * np.abs(b_ref).max()
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref = o.jacobian(U, U1)
    hb.assemble_jacobian()
    x = rng.standard_normal(o.ndof)
    assert np.abs(hb.spmv(x) - A_ref @ x).max() <= 1e-10 * np.abs(A_ref @ x).max()
    hb.close()


@pytest.mark.parametrize("case_name", ["cylinder", "mooney_rivlin", "robin"])
def test_mfma_jacobian_kernel_matches_the_complex_step_oracle(case_name, tmp_path):
    """Row N1 under the driver's eyes: the context is created through ``fsi_create_tuned`` with
    ``FsiTuning.jacobian_mfma = 1``, so the refresh runs ``k_jacobian_mfma`` (v-equation contraction on
    ``v_mfma_f64_16x16x4_f64``), and the ASSEMBLED MATRIX is compared entry by entry with the oracle's complex-step Jacobian at
    the bound of ``test_jacobian_spmv_and_solve_match_oracle`` - on the cylinder (StVK), the predeform problem (MooneyRivlin +
    Robin, theta = 1) and the cylinder with an aneurysm-style Robin wall; the vector-pipe kernel of a second context must give
    the same matrix to round-off."""
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    if case_name == "mooney_rivlin":
        case = prepare_case("predeform", GOLDEN / "cylinder" / "cylinder.h5", tmp_path, dt="0.01", T="0.02", theta="1.0")
    else:
        case = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tmp_path)
    ns, desc = case[0], dict(case[1])
    mesh = ns["mesh"]
    if case_name == "robin":
        fids = np.nonzero(ns["boundaries"] == 33)[0]
        desc["robin_facets"] = mesh.facet_nodes[fids]
        desc["robin_k"] = np.full(len(fids), 1e5)
        desc["robin_c"] = np.full(len(fids), 10.0)
    o = FsiOracle(desc)
    U, U1 = random_state(mesh, o.ndof, seed=11)
    g, P = boundary_data(case, 0.05)
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref = o.jacobian(U, U1)
    rowmax = np.maximum(np.abs(A_ref).max(axis=1).toarray().ravel(), 1e-300)
    mats = {}
    for mfma in (1, 0):
        hb = HipBackend(desc, tuning=dict(jacobian_mfma=mfma))
        assert hb.tuning()["jacobian_mfma"] == mfma                        # the context really runs the kernel asked for
        hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hb.assemble_residual()
        hb.assemble_jacobian()
        mats[mfma] = hb.matrix()
        hb.close()
    rel = np.abs(mats[1] - A_ref).max(axis=1).toarray().ravel() / rowmax
    bound = 1e-10 if case_name == "mooney_rivlin" else 1e-11                # (the product-level bounds of the tests above)
    assert rel.max() < bound, (case_name, rel.max())
    both = np.abs(mats[1] - mats[0]).max(axis=1).toarray().ravel() / rowmax
    assert both.max() < 1e-13, (case_name, both.max())                     # two summation orders of the same contraction


@pytest.mark.parametrize("case_name", ["cylinder", "mooney_rivlin", "robin"])
def test_displacement_rows_in_pair_form_give_the_same_products(case_name, tmp_path):
    """Round 5: the outer product takes the three displacement rows of a node from a pair form (``FsiTuning.compact_drows``: [dd_ii,
    dv_ii] per node pair, extracted and CHECKED at every refresh).  Here: the library adopts the form on the StVK, the MooneyRivlin +
    Robin and the Robin-wall cases (``sweep_flags`` bit 6), the assembled matrix really has nothing else in those rows (checked on the
    host from ``fsi_get_matrix``, i.e. from the other side), and the product equals the one of a context that streams all six value
    rows (to round-off: the sums run in another order) and the oracle's."""
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    if case_name == "mooney_rivlin":
        case = prepare_case("predeform", GOLDEN / "cylinder" / "cylinder.h5", tmp_path, dt="0.01", T="0.02", theta="1.0")
    else:
        case = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tmp_path)
    ns, desc = case[0], dict(case[1])
    mesh = ns["mesh"]
    if case_name == "robin":
        fids = np.nonzero(ns["boundaries"] == 33)[0]
        desc["robin_facets"] = mesh.facet_nodes[fids]
        desc["robin_k"] = np.full(len(fids), 1e5)
        desc["robin_c"] = np.full(len(fids), 10.0)
    o = FsiOracle(desc)
    U, U1 = random_state(mesh, o.ndof, seed=23)
    g, P = boundary_data(case, 0.05)
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref = o.jacobian(U, U1)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(o.ndof)
    ys, A = {}, None
    for compact in (1, 0):
        hb = HipBackend(desc, tuning=dict(compact_drows=compact))
        assert hb.tuning()["compact_drows"] == compact
        hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hb.assemble_residual()
        hb.assemble_jacobian()
        assert bool(int(hb.timers()["sweep_flags"]) & 64) == bool(compact)      # adopted by the refresh's own check / not asked for
        ys[compact] = hb.spmv(x)
        if compact:
            A = hb.matrix().tocsr()
        hb.close()
    # the d rows of the assembled matrix from the host's side: row (node, i) has entries in the columns d_i and v_i of its neighbours only
    nd = 3 * mesh.num_nodes
    Ad = A[:nd].tocoo()
    keep = Ad.data != 0.0
    rows, cols = Ad.row[keep], Ad.col[keep]
    assert cols.max() < 2 * nd                                                   # no pressure column
    assert np.array_equal(rows % 3, cols % 3)                                    # [d | v | p] numbering, three components per node
    ref = A_ref @ x
    scale = np.abs(ref).max()
    assert np.abs(ys[1] - ys[0]).max() <= 1e-13 * scale, np.abs(ys[1] - ys[0]).max() / scale
    assert np.abs(ys[1] - ref).max() <= 1e-10 * scale


def test_a_displacement_row_with_another_entry_keeps_the_full_product(tmp_path, monkeypatch):
    """The other side of ``compact_drows``: the pair form is adopted only if the refresh's check finds nothing else in the d rows.
    ``FSI_DEBUG_DROWS_INJECT`` (a test hook) puts a value where the forms leave a structural zero (a d_y column in a d_x row)
    into the assembled matrix; the library must notice (``sweep_flags`` bit 6 off) and its product must be the product of THAT
    matrix, entry included."""
    from vasp_amd.capi import HipBackend
    case = prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tmp_path)
    ns, desc = case[0], dict(case[1])
    mesh = ns["mesh"]
    U, U1 = random_state(mesh, mesh.num_dofs, seed=3)
    g, P = boundary_data(case, 0.05)
    x = np.random.default_rng(8).standard_normal(mesh.num_dofs)
    out = {}
    for inject in (0, 1):
        if inject:
            monkeypatch.setenv("FSI_DEBUG_DROWS_INJECT", "1")
        hb = HipBackend(desc)
        hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hb.assemble_residual()
        hb.assemble_jacobian()
        out[inject] = (bool(int(hb.timers()["sweep_flags"]) & 64), hb.spmv(x), hb.matrix().tocsr())
        hb.close()
    assert out[0][0] and not out[1][0]
    dA = (out[1][2] - out[0][2]).tocoo()
    dA.eliminate_zeros()
    assert dA.nnz == 1 and dA.row[0] < 3 * mesh.num_nodes and dA.col[0] < 3 * mesh.num_nodes and dA.row[0] % 3 != dA.col[0] % 3
    for inject in (0, 1):
        ref = out[inject][2] @ x
        assert np.abs(out[inject][1] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(out[1][1] - out[0][1]).max() > 0.0          # the entry is in the product


def test_displacement_smoothing_is_chosen_by_context_size(cylinder_case):
    """``FsiTuning.mg_post = 0`` (the default) is resolved when the context is created - seven sweeps after the coarse correction of
    the displacement cycle below 1.1 M P2 nodes, five above (profiles/r05_param_scan_final_tree.txt) - and ``fsi_get_tuning`` returns
    what was taken; an explicit value is kept."""
    from vasp_amd.capi import HipBackend, FsiTuning, load_library
    import ctypes
    t = FsiTuning()
    load_library().fsi_tuning_defaults(ctypes.byref(t))
    assert t.mg_post == 0
    desc = cylinder_case[1]
    hb = HipBackend(desc)
    assert hb.tuning()["mg_post"] == 7
    hb.close()
    hb = HipBackend(desc, tuning=dict(mg_post=5))
    assert hb.tuning()["mg_post"] == 5
    hb.close()


def test_predeform_runs(tmp_path):
    """REF tests/test_simulations.py:60-77: the predeform problem runs a few steps; printed flow properties are sane."""
    import re
    from vasp_amd import monolithic
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns = monolithic.run(["-p", "predeform", "-dt", "0.01", "-T", "0.03", "--folder", str(tmp_path), "--sub-folder", "1",
                             "--new-arguments", f"mesh_path={GOLDEN / 'cylinder' / 'cylinder.h5'}"], out=print)
    out = buf.getvalue()
    vals = np.array([[float(x) for x in row] for row in re.findall(r"Velocity \(mean, min, max\): (.*), (.*), (.*)", out)])
    assert len(vals) == 4 and np.all(np.isfinite(vals)) and np.all(vals >= 0)
    assert out.count("Solved for timestep") == 4 and ns["theta"] == 1.0


def test_offset_stenosis_five_steps_match_converged_golden(stenosis_case):
    """The reference's known-answer case converged to 1e-11 on both sides: HIP path vs the oracle's committed run
    (58 611 dofs, 5 steps, dt 0.01): velocity, displacement and pressure fields to 1e-6 relative."""
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = stenosis_case
    mesh = ns["mesh"]
    gold = np.load(GOLDEN / "stenosis_tight.npz")["states"]
    hb = HipBackend(desc, lin_rtol=1e-11)
    N2 = mesh.num_nodes
    for k in range(5):
        g, P = boundary_data(stenosis_case, 0.01 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=1e-11, rtol=1e-14, max_it=50, lmbda=1.0, recompute=20,
                               recompute_tstep=20)
        # 1e-11 is the round-off floor of |b| on this case (rows of the 1e7 penalty): the loop leaves either below it or on an
        # update norm below 1e-14 with |b| a few per cent above it (observed 1.06e-11 with one preconditioner configuration)
        assert hist[-1][0] < 2e-11
        hb.shift()
    U = hb.get_state("n")
    for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
        err = np.linalg.norm(U[sl] - gold[4][sl]) / np.linalg.norm(gold[4][sl])
        assert err < 1e-6, (name, err)
    hb.close()


def test_device_diagnostics_match_host(cylinder_case):
    """fsi_probe / fsi_flow_stats (post_solve diagnostics on the device) vs the numpy versions of simulation_common."""
    from vasp_amd.capi import HipBackend
    from vasp_amd.fem import MixedFunction
    from vasp_amd.simulation_common import dg0_jacobian, dg0_velocity_magnitude
    ns, desc = cylinder_case[0], cylinder_case[1]
    mesh = ns["mesh"]
    U = np.load(GOLDEN / "cylinder_tight.npz")["states"][2].copy()
    U[:3 * mesh.num_nodes] *= 1e3                                     # make det(I + grad d) visibly different from 1
    hb = HipBackend(desc)
    hb.set_state("n", U)
    d, v, p = mesh.split(U)
    vm = dg0_velocity_magnitude(mesh, v)
    jm = dg0_jacobian(mesh, d)
    mean, mn, mx, minj = hb.flow_stats()
    assert np.isclose(mean, vm.mean(), rtol=1e-12) and np.isclose(mn, vm.min(), rtol=1e-10, atol=1e-300)
    assert np.isclose(mx, vm.max(), rtol=1e-12) and np.isclose(minj, jm.min(), rtol=1e-12)
    pts = mesh.cell_midpoints()[::101] + 1e-6
    cells, bary = mesh.locate(pts)
    out = hb.probe(cells, bary)
    f = MixedFunction(mesh, U)
    for i, x in enumerate(pts):
        assert np.allclose(out[i, 0:3], f.sub(0)(x), rtol=1e-12, atol=1e-300)
        assert np.allclose(out[i, 3:6], f.sub(1)(x), rtol=1e-12, atol=1e-300)
        assert np.isclose(out[i, 6], f.sub(2)(x), rtol=1e-12, atol=1e-300)
    hb.close()


def test_two_level_displacement_solve_is_a_preconditioner_only(stenosis_case, monkeypatch):
    """The P2 -> P1 two-level solve of the displacement block (Galerkin coarse operator on the vertex graph) replaces 60
    one-level Chebyshev sweeps inside the block preconditioner: the converged fields must not move (1e-8), and the outer
    Krylov iteration counts must stay in the same range."""
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = stenosis_case
    out = {}
    for mg in ("1", "0"):
        monkeypatch.setenv("FSI_DD_MG", mg)
        # three decades above the round-off floor of this case (|b| ~ 1e-11, see the five-step golden test): near the floor
        # the Krylov counts are decided by stagnation restarts and by the summation order of the atomics (387 in one process,
        # 620 in another, for the same two-level run at 1e-10), not by the preconditioner this test compares
        hb = HipBackend(desc, lin_rtol=1e-8)
        its = 0
        for k in range(2):
            g, P = boundary_data(stenosis_case, 0.01 * (k + 1))
            hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
            hist = hb.newton_solve(counter=k, first_step_num=0, atol=1e-8, rtol=1e-14, max_it=50, lmbda=1.0, recompute=20,
                                   recompute_tstep=20)
            assert hist[-1][0] < 1e-8
            its += sum(h[3] for h in hist)
            hb.shift()
        out[mg] = (hb.get_state("n"), its)
        hb.close()
    (x1, its1), (x0, its0) = out["1"], out["0"]
    print("krylov iterations: two-level", its1, "one-level", its0)
    mesh = ns["mesh"]
    for name, a, b in zip("dvp", mesh.split(x1), mesh.split(x0)):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), name       # both runs stop on |b| < 1e-8 of a quasi-Newton loop
    assert its1 <= 1.25 * its0


def test_block_preconditioner_is_a_fixed_linear_operator_close_to_the_inverse(stenosis_case):
    """fsi_apply_preconditioner: one application of the field-split block preconditioner as the Krylov method applies it.  Fixed
    sweep counts make it a LINEAR operator (what a flexible method would not need, GCR with recycled directions does: the kept
    pairs q = A M^-1 ... stay pairs only if M^-1 is the same map every time) - to the rounding of its FP32 / FP16 sweeps; and it
    approximates the inverse: the spectrum of A M^-1 sampled by an Arnoldi process sits around one, the same vector comes back
    from two calls bit for bit, and an unassembled context refuses."""
    from vasp_amd.capi import HipBackend
    ns, desc, bc_values, pressure, hook = stenosis_case
    hb = HipBackend(desc)
    with pytest.raises(RuntimeError):
        hb.apply_preconditioner(np.zeros(hb.ndof))                   # no Jacobian yet
    g, P = boundary_data(stenosis_case, 0.01)
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.newton_solve(counter=0, first_step_num=0, atol=1e-6, rtol=1e-6, max_it=50, lmbda=1.0, recompute=20, recompute_tstep=20)
    hb.shift()
    g, P = boundary_data(stenosis_case, 0.02)
    hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_jacobian(); hb.assemble_residual()
    b = hb.get_state("b")
    rng = np.random.default_rng(3)
    bc = np.zeros(hb.ndof, bool); bc[np.asarray(desc["bc_dofs"])] = True
    x = np.where(bc, 0.0, rng.standard_normal(hb.ndof)) * np.abs(b).max()
    y = np.where(bc, 0.0, rng.standard_normal(hb.ndof)) * np.abs(b).max()
    mx, my, mb = hb.apply_preconditioner(x), hb.apply_preconditioner(y), hb.apply_preconditioner(b)
    assert np.array_equal(mx, hb.apply_preconditioner(x))            # the same map, bit for bit
    mxy = hb.apply_preconditioner(2.0 * x - 3.0 * y + b)
    lin = 2.0 * mx - 3.0 * my + mb
    mesh = ns["mesh"]
    for name, u, v in zip("dvp", mesh.split(mxy), mesh.split(lin)):
        assert np.linalg.norm(u - v) <= 2e-3 * np.linalg.norm(v), (name, np.linalg.norm(u - v) / np.linalg.norm(v))
    # Arnoldi on A M^-1 from the right-hand side: Ritz values inside (0, 2) - no outlier a Chebyshev interval missed, no sign change
    q = [b / np.linalg.norm(b)]
    K = 12
    H = np.zeros((K + 1, K))
    for k in range(K):
        w = hb.spmv(hb.apply_preconditioner(q[k]))
        for _ in range(2):
            for j in range(k + 1):
                c = q[j] @ w
                H[j, k] += c
                w -= c * q[j]
        H[k + 1, k] = np.linalg.norm(w)
        q.append(w / H[k + 1, k])
    ritz = np.linalg.eigvals(H[:K, :K])
    print("Ritz values of A M^-1:", np.sort_complex(ritz))
    assert ritz.real.min() > 0.0 and np.abs(ritz).max() < 2.0, ritz
    hb.close()


def test_avf_two_mooney_rivlin_regions_match_oracle(tmp_path):
    """BASELINE config 4's problem file on the HIP path [REF src/vasp/simulations/avf.py:55-84,189-215]: two MooneyRivlin
    regions, Robin condition on both outer walls, the pressure term on both dS(fsi_id[k]).  Residual and Jacobian against
    the oracle at a rough state, then three time steps whose results must solve the ORACLE's discrete equations."""
    from conftest import make_avf_case
    from oracle.fsi_oracle import FsiOracle
    from vasp_amd.capi import HipBackend
    case = make_avf_case(tmp_path)
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    assert desc["solid_models"] == [1, 1] and len(desc["robin_facets"]) > 0
    o = FsiOracle(desc)
    hb = HipBackend(desc, lin_rtol=1e-10)
    rng = np.random.default_rng(11)
    N2, h = mesh.num_nodes, mesh.hmin()
    U, U1 = np.zeros(o.ndof), np.zeros(o.ndof)
    U[:3 * N2] = 0.002 * h * rng.standard_normal(3 * N2)     # the generated mesh has slivers: keep det F > 0 everywhere
    U1[:3 * N2] = 0.9 * U[:3 * N2]
    U[3 * N2:6 * N2] = 0.05 * rng.standard_normal(3 * N2)
    U1[3 * N2:6 * N2] = 0.9 * U[3 * N2:6 * N2]
    U[6 * N2:] = rng.standard_normal(mesh.num_vertices)
    g, P = boundary_data(case, 0.1)
    assert P > 0 and np.abs(g).max() > 0
    assert np.all(np.isfinite(o.rhs(U, U1, P, g)))
    hb.set_state("n", U); hb.set_state("n-1", U1); hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
    hb.assemble_residual()
    b_ref = o.rhs(U, U1, P, g)
    assert np.abs(hb.get_state("b") - b_ref).max() <= 1e-11 * np.abs(b_ref).max()
    o.solver_setup(np.zeros(o.ndof), np.zeros(o.ndof))
    A_ref = o.jacobian(U, U1)
    hb.assemble_jacobian()
    x = rng.standard_normal(o.ndof)
    assert np.abs(hb.spmv(x) - A_ref @ x).max() <= 1e-10 * np.abs(A_ref @ x).max()
    # three time steps from rest, Newton driven well below the problem's own tolerance
    Z = np.zeros(o.ndof)
    hb.set_state("n", Z); hb.set_state("n-1", Z)
    prev = Z.copy()
    for k in range(3):
        g, P = boundary_data(case, 1e-4 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=1e-10, rtol=1e-14, max_it=30, lmbda=1.0, recompute=30,
                               recompute_tstep=10)
        Uk = hb.get_state("n")
        b = o.rhs(Uk, prev, P, g)                       # the oracle's residual at the HIP path's solution
        b0 = o.rhs(prev, prev, P, g)
        # (what the loop guarantees is |b| < atol; 1e-7 |b0| is ~3e-13 on the first steps from rest, which the mixed default reaches and
        # an all-FP64 basis misses by a factor two: both are converged runs)
        assert np.linalg.norm(b) <= max(1e-7 * np.linalg.norm(b0), 1e-11), (k, np.linalg.norm(b), np.linalg.norm(b0), hist)
        assert np.all(np.isfinite(Uk))
        hb.shift()
        prev = Uk
    hb.close()


def test_aneurysm_three_steps_match_converged_golden(tmp_path):
    """BASELINE configs 3/5's problem file (Robin wall) on the fixture the reference tests it with [REF
    tests/test_simulations.py:80-90, inlet_id=4]: the reference only checks finiteness there; here the three steps are
    compared field by field with the oracle's committed converged run (tests/golden/aneurysm_tight.npz)."""
    from vasp_amd.capi import HipBackend
    gold = GOLDEN / "aneurysm_tight.npz"
    if not gold.exists():
        pytest.skip("aneurysm_tight.npz not generated")
    G = np.load(gold)["states"]
    case = prepare_case("aneurysm", GOLDEN / "aneurysm" / "small_aneurysm.h5", tmp_path, extra=("inlet_id=4",))
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    hb = HipBackend(desc, lin_rtol=1e-10)
    N2 = mesh.num_nodes
    for k in range(3):
        g, P = boundary_data(case, 1e-3 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hb.newton_solve(counter=k, first_step_num=0, atol=1e-11, rtol=1e-14, max_it=30, lmbda=1.0, recompute=ns["recompute"],
                        recompute_tstep=ns["recompute_tstep"])
        hb.shift()
        U = hb.get_state("n")
        for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
            err = np.linalg.norm(U[sl] - G[k][sl]) / np.linalg.norm(G[k][sl])
            assert err < 1e-6, (k, name, err)
    hb.close()


# BASELINE configs[4] ("mixed FP64 assembly / FP32 Krylov" on the aneurysm / Robin workload) against the oracle's converged
# run, per field (bound = 2x the value measured on MI355X in round 3; the values are in DESIGN.md section 2):
#   "own":   the problem file's own tolerances (atol 1e-10, rtol 1e-9 [REF src/vasp/simulations/aneurysm.py:48-49]); the
#            storage policy decides (these tolerances select the FP64 basis);
#   "mixed": tolerances 1e-6 (the other problem files' choice), for which the policy selects the FP32 Krylov basis, the FP32
#            Jacobian copy inside the iterations and FP16 preconditioner matrices.
CONFIG5_BOUND = {"own": {"d": 5e-6, "v": 1e-6, "p": 1e-7},      # measured 2.1e-6, 3.0e-7, 1.3e-8 (the loop stops on the update norm 1e-9)
                 "mixed": {"d": 2e-4, "v": 2e-4, "p": 2e-4}}


@pytest.mark.parametrize("mode", ["own", "mixed"])
def test_aneurysm_production_defaults_against_the_converged_oracle_run(tmp_path, mode):
    """the production defaults (storage chosen by the policy, forcing 1e-2) had never been
    compared with the oracle on the aneurysm / Robin workload - the golden test above drives Newton to 1e-11 with every
    linear system solved to 1e-10.  Three steps with the problem file's Jacobian policy; the states must stay within the
    Newton stopping tolerance of the oracle's converged states (tests/golden/aneurysm_tight.npz)."""
    from vasp_amd.capi import HipBackend
    G = np.load(GOLDEN / "aneurysm_tight.npz")["states"]
    case = prepare_case("aneurysm", GOLDEN / "aneurysm" / "small_aneurysm.h5", tmp_path, extra=("inlet_id=4",))
    ns, desc = case[0], case[1]
    mesh = ns["mesh"]
    atol, rtol = (ns["atol"], ns["rtol"]) if mode == "own" else (1e-6, 1e-6)
    hb = HipBackend(desc)                                                    # production defaults
    N2 = mesh.num_nodes
    worst, its, kry = {"d": 0.0, "v": 0.0, "p": 0.0}, [], 0
    for k in range(3):
        g, P = boundary_data(case, 1e-3 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        h = hb.newton_solve(counter=k, first_step_num=0, atol=atol, rtol=rtol, max_it=ns["max_it"], lmbda=1.0,
                            recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
        its.append(len(h)); kry += sum(it[3] for it in h)
        hb.shift()
        U = hb.get_state("n")
        assert np.isfinite(U).all()
        for name, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None))):
            worst[name] = max(worst[name], float(np.linalg.norm(U[sl] - G[k][sl]) / np.linalg.norm(G[k][sl])))
    tm = hb.timers()
    hb.close()
    print(f"aneurysm [{mode}] atol {atol:g} rtol {rtol:g}, Q in FP{8 * tm['q_elem_bytes']}: distance to the converged oracle states",
          worst, "Newton", its, "Krylov", kry)
    with contextlib.suppress(OSError):
        import json
        from conftest import ROOT
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / f"config5_aneurysm_{mode}.json").write_text(json.dumps(
            dict(errors=worst, newton=its, krylov=kry, q_elem_bytes=tm["q_elem_bytes"], spmv_fp32_calls=tm["spmv_fp32_calls"],
                 atol=atol, rtol=rtol)))
    if mode == "mixed":
        assert tm["q_elem_bytes"] == 4 and tm["spmv_fp32_calls"] > 0          # the mixed mode really ran
    for name, bound in CONFIG5_BOUND[mode].items():
        assert worst[name] < bound, (name, worst[name], bound)


def test_stress_strain_kernel_matches_oracle(cyl, cylinder_case):
    """fsi_stress_strain vs oracle/post_oracle.py on every solid cell of the cylinder at a strained state."""
    from oracle.post_oracle import stress_strain_dg1
    ns, desc = cylinder_case[0], cylinder_case[1]
    mesh = ns["mesh"]
    rng = np.random.default_rng(5)
    U = np.zeros(cyl.ndof)
    N2 = mesh.num_nodes
    U[:3 * N2] = 0.03 * mesh.hmin() * rng.standard_normal(3 * N2)
    cyl.set_state("n", U)
    solid = np.nonzero(np.asarray(desc["cell_kind"]) == 1)[0]
    got = cyl.stress_strain(solid)
    dn = U[:3 * N2].reshape(N2, 3)
    ref = stress_strain_dg1(mesh.coords, mesh.tets, mesh.tet_nodes, dn, solid, desc["solid_props"][0], eig="kopp")
    lap = stress_strain_dg1(mesh.coords, mesh.tets, mesh.tet_nodes, dn, solid, desc["solid_props"][0])
    for key in ("TrueStress", "GreenLagrangeStrain", "MaxPrincipalStress", "MaxPrincipalStrain"):
        scale = np.abs(ref[key]).max()
        # tensors: same arithmetic, round-off; principal values: the kernel uses get_eig's closed form, whose own round-off
        # is ~1e-8 of the tensor (acos of a ratio near 1), so it is held to the same formula at 1e-7 and to LAPACK at 1e-6
        tol = 1e-10 if key in ("TrueStress", "GreenLagrangeStrain") else 1e-7
        assert np.abs(got[key] - ref[key]).max() <= tol * scale, key
        assert np.abs(got[key] - lap[key]).max() <= 1e-6 * scale, key
    with pytest.raises(Exception):
        cyl.stress_strain(np.nonzero(np.asarray(desc["cell_kind"]) == 0)[0][:2])       # fluid cells are refused
    cyl.set_state("n", np.zeros(cyl.ndof))


def test_wall_shear_stress_kernel_matches_oracle(cyl, cylinder_case):
    """fsi_wall_shear_stress vs oracle/post_oracle.py on the exterior facets of the fluid sub-mesh (cells with one and
    with several boundary facets)."""
    from oracle.post_oracle import wall_shear_stress
    from test_post_oracle import fluid_boundary_facets
    ns, desc = cylinder_case[0], cylinder_case[1]
    mesh = ns["mesh"]
    rng = np.random.default_rng(6)
    U = np.zeros(cyl.ndof)
    N2 = mesh.num_nodes
    U[3 * N2:6 * N2] = rng.standard_normal(3 * N2)
    cyl.set_state("n", U)
    fids, cell, local = fluid_boundary_facets(mesh)
    assert (np.bincount(cell)[cell] > 1).any()                      # corner cells with two boundary facets are covered
    mu = 3.5e-3
    got = cyl.wall_shear_stress(cell, local, mu)
    ref = wall_shear_stress(mesh.coords, mesh.tets, mesh.tet_nodes, U[3 * N2:6 * N2].reshape(N2, 3), cell, local, mu)
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()
    cyl.set_state("n", np.zeros(cyl.ndof))


def _three_steps(case, recompute_tstep=2):
    """Three time steps of the production policy in a fresh context; returns (residual vector of the first assembly, product
    of the first Jacobian with a fixed vector, state after every step, Krylov iterations)."""
    from vasp_amd.capi import HipBackend
    hb = HipBackend(case[1])
    x = np.random.default_rng(7).standard_normal(hb.ndof)
    states, its = [], []
    b0 = ax0 = None
    for k in range(3):
        g, P = boundary_data(case, 1e-3 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        if k == 0:
            hb.assemble_residual()
            b0 = hb.get_state("b")
            hb.assemble_jacobian()
            ax0 = hb.spmv(x)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=1e-6, rtol=1e-6, max_it=20, lmbda=1.0, recompute=20,
                               recompute_tstep=recompute_tstep)
        its.append([h[3] for h in hist])
        hb.shift()
        states.append(hb.get_state("n"))
    hb.close()
    return b0, ax0, states, its




def test_time_steps_are_bitwise_reproducible(tmp_path):
    """`north_star`: segmented scatter-add into the global vector / CSR matrix.  Here the residual's segmented reduction runs
    on the owner's side (element vectors stored per cell, every dof sums the cells around its node in ascending order); for
    the Jacobian the cells are coloured (no two cells of a colour share a node) and every colour is one launch, so no entry
    of A receives two adds whose order could vary; the small facet scatters are merged on the host, the norms and the coarse
    operators of the preconditioner are summed in fixed orders.  Two fresh contexts therefore produce THE SAME BITS -
    residual, Jacobian, Krylov iteration counts and the state after three steps with two Jacobian refreshes - on a mesh
    large enough (50 k tets, 4 000 workgroups in flight) for unordered atomics to show."""
    from vasp_amd.meshgen import write_mesh
    write_mesh(tmp_path / "s.h5", 50000)
    case = prepare_case("offset_stenosis", tmp_path / "s.h5", tmp_path / "run", dt="0.001", T="0.002")
    b_a, ax_a, st_a, it_a = _three_steps(case)
    b_b, ax_b, st_b, it_b = _three_steps(case)
    assert np.array_equal(b_a, b_b)
    assert np.array_equal(ax_a, ax_b)
    assert it_a == it_b
    for u, w in zip(st_a, st_b):
        assert np.array_equal(u, w)


def test_reproducible_assembly_equals_the_atomic_one_to_roundoff(tmp_path, monkeypatch):
    """FSI_ASSEMBLY=atomic (one launch over all cells, unordered atomics) against the default (gathered residual, coloured
    Jacobian): the same sums in another order - residual and Jacobian product agree to round-off, not to the bit."""
    from vasp_amd.capi import HipBackend
    from vasp_amd.meshgen import write_mesh
    write_mesh(tmp_path / "s.h5", 50000)
    case = prepare_case("offset_stenosis", tmp_path / "s.h5", tmp_path / "run", dt="0.001", T="0.002")
    mesh = case[0]["mesh"]
    out = {}
    for mode in ("default", "atomic"):
        if mode == "atomic":
            monkeypatch.setenv("FSI_ASSEMBLY", "atomic")
        else:
            monkeypatch.delenv("FSI_ASSEMBLY", raising=False)
        hb = HipBackend(case[1])
        assert (hb.timers()["assembly_colours"] > 0) == (mode == "default")
        U, U1 = random_state(mesh, hb.ndof, seed=3)
        g, P = boundary_data(case, 1e-3)
        hb.set_state("n", U); hb.set_state("n-1", U1)
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hb.assemble_residual()
        b = hb.get_state("b")
        hb.assemble_jacobian()
        out[mode] = (b, hb.spmv(np.random.default_rng(5).standard_normal(hb.ndof)))
        hb.close()
    for k in range(2):
        assert np.abs(out["default"][k] - out["atomic"][k]).max() <= 1e-13 * np.abs(out["atomic"][k]).max()


def test_kept_basis_stays_orthonormal_on_the_avf_problem(tmp_path):
    """Regression for the loss of orthogonality of the recycled Krylov basis: 25 steps of the avf
    problem file (dt = 1e-4, two Jacobian lifetimes and a half) with the production tolerances.  With the second
    Gram-Schmidt pass decided by the tolerance alone (round 2) this run needed 807 Krylov iterations, 117 Arnoldi steps out
    of stagnant residuals and lost the FP32 basis once; with the criterion the pass itself supplies: 605, none, none."""
    from conftest import make_avf_case
    from vasp_amd.capi import HipBackend
    case = make_avf_case(tmp_path)
    ns, desc = case[0], case[1]
    hb = HipBackend(desc)
    krylov = 0
    for k in range(25):
        g, P = boundary_data(case, 1e-4 * (k + 1))
        hb.set_dirichlet_values(g); hb.set_interface_pressure(P)
        hist = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=50, lmbda=1.0,
                               recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
        assert hist[-1][0] < ns["atol"] or hist[-1][1] < ns["rtol"]
        krylov += sum(h[3] for h in hist)
        hb.shift()
    tm = hb.timers()
    hb.close()
    assert tm["fp32_fallbacks"] == 0 and tm["gcr_restarts"] == 0 and tm["newton_retries"] == 0
    assert tm["gcr_arnoldi_steps"] <= 10 and krylov <= 700, (tm["gcr_arnoldi_steps"], krylov)
