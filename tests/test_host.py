"""Host side of the boundary: CLI, parameter handling, hooks, boundary data and log grammar of the re-hosted problem
files (mirrors what VaSP's own tests observe through stdout; REF tests/test_simulations.py, log_plotter.py:71-82)."""
import contextlib
import io
import re

import numpy as np
import pytest

from conftest import GOLDEN
from vasp_amd import monolithic
from vasp_amd.simulation_common import InterfacePressure

# the reference's log grammar [REF src/vasp/postprocessing/log_plotter.py:71-81]
TIME_STEP = re.compile(r"Solved for timestep (.*), t = (.*) in (.*) s")
RAMP = re.compile(r"ramp_factor = (.*) m\^3/s")
PRESSURE = re.compile(r"Instantaneous normal stress prescribed at the FSI interface (.*) Pa")
NEWTON = re.compile(r'Newton iteration (.*): r \(atol\) = (.*) \(tol = .*\), r \(rel\) = (.*) \(tol = .*\)')
PROBE = re.compile(r"Probe Point (.*): Velocity: \((.*), (.*), (.*)\) \| Pressure: (.*)")
PROBE_D = re.compile(r"Probe Point (.*): Displacement: \((.*), (.*), (.*)\)")
FLOW = re.compile(r"\s*Flow Rate at Inlet: (.*)")
VEL = re.compile(r"\s*Velocity \(mean, min, max\): (.*), (.*), (.*)")
CFL = re.compile(r"\s*CFL \(mean, min, max\): (.*), (.*), (.*)")
RE = re.compile(r"\s*Reynolds Numbers \(mean, min, max\): (.*), (.*), (.*)")


def test_cli_matches_the_reference_invocation():
    # REF tests/test_simulations.py:22-23
    a = monolithic.parse("-p offset_stenosis -dt 0.01 -T 0.04 --verbose True --theta 0.51 --folder tmp --sub-folder 1 "
                         "--new-arguments mesh_path=some/mesh.h5 inlet_id=4".split())
    assert a == dict(problem="offset_stenosis", dt=0.01, T=0.04, verbose=True, theta=0.51, folder="tmp", sub_folder="1",
                     mesh_path="some/mesh.h5", inlet_id=4)
    with pytest.raises(SystemExit):
        monolithic.parse(["--new-arguments", "novalue"])


def test_problem_parameters_offset_stenosis():
    # REF src/vasp/simulations/offset_stenosis.py:27-82
    from vasp_amd.problems import default_variables, offset_stenosis
    v = offset_stenosis.set_problem_parameters(dict(default_variables))
    assert v["mu_s"] == pytest.approx(1e6 / 2.9) and v["lambda_s"] == pytest.approx(0.45 * 2 * (1e6 / 2.9) / 0.1)
    assert (v["atol"], v["rtol"], v["recompute"], v["recompute_tstep"]) == (1e-6, 1e-6, 20, 20)
    assert v["dx_f_id"] == [1, 1001] and v["mu_f"] == [1.5e-3, 1.0e-2] and v["fsi_region"] == [0.008, 0, 0, 0.008]
    # key set / order of turtleFSI's default_variables [REF tests/test_data/hemodynamics_data/Checkpoint/default_variables.json]
    keys = list(default_variables)
    assert keys[:8] == ["dt", "theta", "T", "t", "counter", "v_deg", "p_deg", "d_deg"]
    assert keys.index("atol") < keys.index("rtol") < keys.index("max_it") < keys.index("lmbda") < keys.index("recompute")
    monolithic.build_properties(v)
    assert v["fluid_properties"] == [dict(dx_f_id=1, rho_f=1e3, mu_f=1.5e-3), dict(dx_f_id=1001, rho_f=1e3, mu_f=1e-2)]
    assert v["solid_properties"][0]["material_model"] == "StVenantKirchoff"


def test_boundary_conditions_offset_stenosis(stenosis_case):
    ns, desc, bc_values, pressure, hook = stenosis_case
    mesh = ns["mesh"]
    N2 = mesh.num_nodes
    bcs = ns["bcs"]
    assert [(b.space.field, b.space.comp, b.marker) for b in bcs] == \
        [(1, 0, 3), (1, 1, 3), (1, 2, 3), (0, None, 3), (1, None, 11), (0, None, 11), (0, None, 11)]  # REF :170-179
    dofs = desc["bc_dofs"]
    assert len(np.unique(dofs)) == len(dofs) and dofs.max() < 6 * N2          # no pressure Dirichlet conditions
    # '+' side of the interface facets is the solid cell [REF :189]
    assert np.all(ns["domains"][desc["pressure_facet_cell"]] == 2) and len(desc["pressure_facets"]) == 290
    # inlet flux of the Womersley profile equals the prescribed flow rate Q(t) (ramped), up to the P2 interpolation
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 0.3
        hook("pre_solve")(**ns)
    g = np.zeros(mesh.num_dofs)
    g[dofs] = bc_values()
    from vasp_amd.simulation_common import inlet_flux
    An, Bn = np.loadtxt(GOLDEN.parent.parent / "vasp_amd" / "problems" / "FC_MCA_10").T
    w = 2 * np.pi / 0.951
    Q = np.real(np.sum((An - 1j * Bn) * 2.5e-6 * np.exp(1j * np.arange(len(An)) * w * 0.3)))
    flux = -inlet_flux(mesh, mesh.split(g)[1], ns["dsi"])
    assert flux == pytest.approx(Q, rel=2e-2)
    # later conditions win on shared dofs: inlet rim nodes (also on the solid end, marker 11) carry zero velocity
    rim = np.intersect1d(np.unique(mesh.facet_nodes[ns["boundaries"] == 3]), np.unique(mesh.facet_nodes[ns["boundaries"] == 11]))
    assert len(rim) and np.all(mesh.split(g)[1][rim] == 0.0)


def test_interface_pressure_formula():
    # REF src/vasp/simulations/simulation_common.py:371-395
    An, Bn = np.array([1.0, 0.1, -0.05]), np.array([0.0, 0.02, 0.03])
    ip = InterfacePressure(t=0.0, t_ramp_start=0.0, t_ramp_end=0.2, An=An, Bn=Bn, period=0.951, P_mean=11200)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ip.update(0.05)
    ramp = 0.5 - 0.5 * np.cos(np.pi * 0.25)
    Pn = abs(sum((An[i] - 1j * Bn[i]) * np.exp(1j * i * 2 * np.pi / 0.951 * 0.05) for i in range(3)))
    assert ip.P == pytest.approx(ramp * Pn * 11200, rel=1e-14)
    lines = buf.getvalue().splitlines()
    assert RAMP.match(lines[0]) and PRESSURE.match(lines[1])
    with contextlib.redirect_stdout(io.StringIO()):
        ip.update(0.5)
    assert ip.P == pytest.approx(abs(sum((An[i] - 1j * Bn[i]) * np.exp(1j * i * 2 * np.pi / 0.951 * 0.5) for i in range(3))) * 11200)


def test_time_loop_and_log_grammar_with_a_stub_backend(tmp_path):
    """Hook order, number of steps (T/dt + 1, as `while t <= T + dt/10`) and every log line the reference's
    log plotter parses - driven with a backend stub, so no GPU and no oracle is involved."""
    calls = []

    class Stub:
        def __init__(self, desc):
            self.n = 6 * desc["num_nodes"] + len(desc["coords"])
            self.x = np.zeros(self.n)
        def set_dirichlet_values(self, v): calls.append("bc")
        def set_interface_pressure(self, P): calls.append("P")
        def newton_solve(self, *, log=None, atol, rtol, **kw):
            calls.append(("newton", kw["counter"], kw["first_step_num"]))
            if log:
                log("Compute Jacobian matrix")
                log("Newton iteration %d: r (atol) = %.3e (tol = %.3e), r (rel) = %.3e (tol = %.3e) " % (0, 1e-3, atol, 1e-7, rtol))
            return [(1e-3, 1e-7, True)]
        def shift(self): calls.append("shift")
        def get_state(self, which, out): out[:] = 1e-6; return out
        def set_state(self, which, x): calls.append(("set_state", which, float(x[0])))

    lines = []
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns = monolithic.run(["-p", "offset_stenosis", "-dt", "0.01", "-T", "0.02", "--theta", "0.51", "--folder", str(tmp_path),
                             "--sub-folder", "1", "--new-arguments",
                             f"mesh_path={GOLDEN / 'offset_stenosis' / 'offset_stenosis.h5'}"], backend_factory=Stub, out=lines.append)
    assert [c for c in calls if isinstance(c, tuple)] == [("newton", 0, 0), ("newton", 1, 0), ("newton", 2, 0)]
    assert calls[:4] == ["bc", "P", ("newton", 0, 0), "shift"]
    assert ns["counter"] == 3 and ns["newton_iterations"] == 3
    solved = [TIME_STEP.match(l) for l in lines if l.startswith("Solved")]
    assert [int(m.group(1)) for m in solved] == [1, 2, 3] and [float(m.group(2)) for m in solved] == [0.01, 0.02, 0.03]
    assert sum(bool(NEWTON.match(l)) for l in lines) == 3
    out = buf.getvalue().splitlines()
    for pat, count in ((RAMP, 3), (PRESSURE, 3), (PROBE, 3 * 7), (PROBE_D, 3 * 50), (FLOW, 3), (VEL, 3), (CFL, 3), (RE, 3)):
        assert sum(bool(pat.match(l)) for l in out) == count, pat.pattern
    assert sum(l.startswith("Minimum Jacobian: ") for l in out) == 3       # the solver's spelling [REF simulation_common.py:343]
    # output tree [REF docs/offset_stenosis.md:209-228] and the relabelled mesh file the post-processing tools read
    from vasp_amd.h5lite import read_h5
    g = read_h5(tmp_path / "1" / "Mesh" / "mesh.h5")
    vals = np.asarray(g["domains"]["values"].data)
    assert (vals == 1001).sum() == 77 and (tmp_path / "1" / "Checkpoint").is_dir() and (tmp_path / "1" / "Visualization").is_dir()
    # save_step = 1, save_deg = 2: three frames per field on the refined mesh (V + E = 9554 nodes); checkpoint at counter 0
    viz = read_h5(tmp_path / "1" / "Visualization" / "velocity.h5")
    assert sorted(viz["VisualisationVector"]) == ["0", "1", "2"] and viz["VisualisationVector"]["2"].data.shape == (9554, 3)
    import json
    meta = json.loads((tmp_path / "1" / "Checkpoint" / "default_variables.json").read_text())
    assert meta["counter"] == 0 and meta["t"] == 0.01 and meta["fsi_region"] == [0.008, 0, 0, 0.008]
    # restart from that checkpoint: state handed to the backend, time continues from the stored t
    calls.clear()
    lines2 = []
    with contextlib.redirect_stdout(io.StringIO()):
        monolithic.run(["-p", "offset_stenosis", "-dt", "0.01", "-T", "0.03", "--theta", "0.51", "--folder", str(tmp_path),
                        "--sub-folder", "2", "--restart-folder", str(tmp_path / "1"), "--new-arguments",
                        f"mesh_path={GOLDEN / 'offset_stenosis' / 'offset_stenosis.h5'}"], backend_factory=Stub, out=lines2.append)
    assert calls[0] == ("set_state", "n", 1e-6) and calls[1][:2] == ("set_state", "n-1")
    assert [float(TIME_STEP.match(l).group(2)) for l in lines2 if l.startswith("Solved")] == [0.02, 0.03, 0.04]


def test_cylinder_hooks(cylinder_case):
    ns, desc, bc_values, pressure, hook = cylinder_case
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns["t"] = 0.05
        hook("pre_solve")(**ns)
    assert pressure.P == pytest.approx(0.5 * 10000)                       # cosine ramp half way [REF cylinder.py:133-157]
    assert "v (centerline, at inlet) = 0.37" in buf.getvalue() and " m/s" in buf.getvalue()
    g = np.zeros(ns["mesh"].num_dofs)
    g[desc["bc_dofs"]] = bc_values()
    v = ns["mesh"].split(g)[1]
    assert np.abs(v).max() == pytest.approx(0.375, rel=0.05)               # parabola peak at the inlet centre


def test_aneurysm_problem_with_robin_condition(tmp_path):
    # REF src/vasp/simulations/aneurysm.py:29-87 and tests/test_simulations.py:80-90 (inlet_id=4 for the small fixture)
    from conftest import prepare_case
    ns, desc, bc_values, pressure, hook = prepare_case("aneurysm", GOLDEN / "aneurysm" / "small_aneurysm.h5", tmp_path,
                                                       extra=("inlet_id=4",))
    assert ns["robin_bc"] and ns["atol"] == 1e-10 and ns["rtol"] == 1e-9
    assert len(desc["robin_facets"]) == 698 and np.all(desc["robin_k"] == 1e5) and np.all(desc["robin_c"] == 10.0)
    assert len(desc["pressure_facets"]) == 698 and ns["probe_points"].shape == (14, 3)
    lo, hi = ns["mesh"].coords.min(axis=0), ns["mesh"].coords.max(axis=0)
    assert np.all(ns["probe_points"] >= lo - 1e-3) and np.all(ns["probe_points"] <= hi + 1e-3)   # mm -> m [REF :157-158]


def test_predeform_problem_mooney_rivlin(tmp_path):
    # REF src/vasp/simulations/predeform.py:27-89 (theta = 1, lmbda = 0.5, dict-valued MooneyRivlin solid_properties)
    from conftest import prepare_case
    ns, desc, bc_values, pressure, hook = prepare_case("predeform", GOLDEN / "cylinder" / "cylinder.h5", tmp_path,
                                                       dt="0.01", T="0.02", theta="1.0")
    assert ns["lmbda"] == 0.5 and ns["theta"] == 1.0 and ns["save_deg"] == 1
    assert desc["solid_models"] == [1] and desc["solid_props"][0][3:] == (0.0, 0.02e6, 1.8e6)
    assert len(desc["robin_facets"]) > 0 and np.all(desc["robin_k"] == 1e5)
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 0.55
        hook("pre_solve")(**ns)
    assert pressure.P == pytest.approx(0.5 * 11332.4)                    # half-way through the pressure ramp [0.2, 0.9]


def test_avf_problem_two_solid_regions(tmp_path):
    """REF src/vasp/simulations/avf.py: two MooneyRivlin regions, list-valued ids, tabulated inlets and pressure.
    The reference tree holds neither the AVF mesh nor avf.csv, so the case is a synthetic tube whose downstream half
    carries the vein ids (1002 / 1011 / 1022 / 1033)."""
    from conftest import make_avf_case
    ns, desc, bc_values, pressure, hook = make_avf_case(tmp_path)
    mesh = ns["mesh"]
    assert desc["solid_models"] == [1, 1] and desc["solid_props"][1][3:] == (0.0, 0.003e6, 0.538e6)
    assert set(np.unique(desc["cell_region"][desc["cell_kind"] == 1])) == {0, 1}
    b = ns["boundaries"]
    assert (b == 1011).sum() > 0 and (b == 1022).sum() > 0 and (b == 22).sum() > 0       # relabel keeps both regions' ids
    nf = (b == 22).sum() + (b == 1022).sum()
    assert len(desc["pressure_facets"]) == nf                                             # both dS(fsi_id[k]) terms
    assert len(desc["robin_facets"]) == (b == 33).sum() + (b == 1033).sum() and np.all(desc["robin_k"] == 1e5)
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = 0.1
        hook("pre_solve")(**ns)
    ramp = -0.5 * np.cos(np.pi / 0.15 * 0.05) + 0.5
    assert pressure.P == pytest.approx(ns["p_out_bc_val"].interp_P[ns["p_out_bc_val"].number] * ramp)
    g = np.zeros(mesh.num_dofs)
    g[desc["bc_dofs"]] = bc_values()
    assert np.abs(mesh.split(g)[1]).max() > 0                                              # inlets carry the table value


class _StubBackend:
    """Stands in for the time-step kernel where only the host driver is under test."""

    def __init__(self, desc):
        self.n = 6 * int(desc["num_nodes"]) + len(desc["coords"])
        self.U = np.zeros(self.n)
        self.steps = 0

    def set_dirichlet_values(self, v): pass
    def set_interface_pressure(self, P): pass

    def newton_solve(self, **kw):
        self.steps += 1
        self.U[:] = self.steps
        return [(1e-3, 1e-4, False, 3, 1e-9), (1e-8, 1e-9, False, 2, 1e-9)]

    def shift(self): pass

    def get_state(self, which, out=None):
        out[:] = self.U
        return out

    def set_state(self, which, x): self.U[:] = x


def _run_cylinder(tmp_path, extra=(), T="0.02"):
    from vasp_amd import monolithic
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns = monolithic.run(["-p", "cylinder", "-dt", "0.001", "-T", T, "--theta", "0.51", "--folder", str(tmp_path), "--sub-folder",
                             "1", "--save-deg", "1", "--new-arguments", f"mesh_path={GOLDEN / 'cylinder' / 'cylinder.h5'}", *extra],
                            backend_factory=_StubBackend, out=print)
    return ns, buf.getvalue()


def test_killturtle_sentinel_checkpoints_and_stops(tmp_path):
    """turtleFSI's run controls: a ``killturtle`` file in the results folder -> checkpoint, stop."""
    (tmp_path / "1").mkdir(parents=True)
    (tmp_path / "1" / "killturtle").write_text("")
    ns, out = _run_cylinder(tmp_path)
    assert ns["backend"].steps == 1 and "killturtle found" in out and out.count("Solved for timestep") == 1
    assert not (tmp_path / "1" / "killturtle").exists()
    import json
    meta = json.loads((tmp_path / "1" / "Checkpoint" / "default_variables.json").read_text())
    assert meta["counter"] == 0 and abs(meta["t"] - 1e-3) < 1e-12


def test_killtime_stops_the_loop_and_restart_continues_the_series(tmp_path):
    ns, out = _run_cylinder(tmp_path, extra=("killtime=0",))
    assert ns["backend"].steps == 1 and "Reached killtime" in out
    # restart from that folder: same folder, run file 1, counter and time carried over
    ns2, out2 = _run_cylinder(tmp_path / "unused", extra=("--restart-folder", str(tmp_path / "1")), T="0.003")
    assert ns2["results_folder"] == tmp_path / "1"
    assert (tmp_path / "1" / "Visualization" / "velocity_run_1.h5").exists()
    text = (tmp_path / "1" / "Visualization" / "velocity.xdmf").read_text()
    assert "velocity.h5:/VisualisationVector/0" in text and "velocity_run_1.h5:/VisualisationVector/0" in text


def test_host_state_is_fetched_only_when_a_hook_reads_it(tmp_path):
    """post_solve of the cylinder problem needs probes / cell statistics / the inlet patch only: the driver must not pull the
    whole state vector off the device every step (it did in round 1: 74 MB per step at 1 M tets)."""
    calls = []

    class Counting(_StubBackend):
        def get_state(self, which, out=None):
            calls.append(which)
            return super().get_state(which, out)

        def get_values(self, which, dofs):
            return self.U[dofs]

        def flow_stats(self):
            return (0.1, 0.0, 0.2, 1.0)

        def probe(self, cells, bary):
            return np.zeros((len(cells), 7))

    from vasp_amd import monolithic
    with contextlib.redirect_stdout(io.StringIO()):
        monolithic.run(["-p", "cylinder", "-dt", "0.001", "-T", "0.004", "--theta", "0.51", "--folder", str(tmp_path), "--sub-folder", "1",
                        "--save-step", "1000", "--checkpoint-step", "1000", "--new-arguments",
                        f"mesh_path={GOLDEN / 'cylinder' / 'cylinder.h5'}"], backend_factory=Counting, out=print)
    assert calls.count("n") <= 1 and "n-1" not in calls          # only the checkpoint / frame of step 0 reads the vector


def test_solver_events_show_up_in_the_product_log(tmp_path):
    """a refresh-and-retry after a failed linear solve (a policy the reference does not have) must be
    visible to a user of `python -m vasp_amd.monolithic`, not only in bench.py: the driver prints the library's event
    counters whenever one of them has grown during a step - and nothing when they stay at zero."""
    from vasp_amd import monolithic

    class Eventful(_StubBackend):
        def solver_events(self):               # run totals (fsi_get_solver_events): cheap, not affected by timer resets
            return dict(newton_retries=1 if self.steps >= 2 else 0, fp32_fallbacks=0, gcr_restarts=0)

        def timers(self, reset=False):         #  the driver must not resolve the phase timers every step
            raise AssertionError("monolithic.run read the phase timers for its event line")

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ns = monolithic.run(["-p", "cylinder", "-dt", "0.001", "-T", "0.003", "--theta", "0.51", "--folder", str(tmp_path), "--sub-folder",
                             "1", "--new-arguments", f"mesh_path={GOLDEN / 'cylinder' / 'cylinder.h5'}"],
                            backend_factory=Eventful, out=print)
    out = buf.getvalue()
    assert out.count("Linear solver events so far: newton_retries = 1, fp32_fallbacks = 0, gcr_restarts = 0") == 1
    assert out.index("Linear solver events") > out.index("Solved for timestep 1,")          # raised in step 2, reported once
    assert ns["solver_events"]["newton_retries"] == 1
    _, quiet = _run_cylinder(tmp_path / "q", T="0.002")                                        # a backend without event counters: no line
    assert "Linear solver events" not in quiet


def test_config_file_is_read_and_the_command_line_wins(tmp_path):
    """`turtleFSI -p problem -c my_config.config` [REF docs/simulation.md:19-31]: `key = value` lines, option names without
    their dashes or problem-file parameters; precedence file < command line, as ConfigArgParse gives turtleFSI."""
    from vasp_amd.monolithic import parse
    cfg = tmp_path / "my_config.config"
    cfg.write_text("# a comment\ndt = 0.002\nend-time: 0.5\ntheta 0.6\nsave-deg = 2\nverbose = False\n"
                   "mesh_path = some/mesh.h5   ; problem-file key\nfsi_region = [0.0, 1.0, 2.0, 3.5]\n")
    a = parse(["-p", "cylinder", "-c", str(cfg)])
    assert a["dt"] == 0.002 and a["T"] == 0.5 and a["theta"] == 0.6 and a["save_deg"] == 2 and a["verbose"] is False
    assert a["mesh_path"] == "some/mesh.h5" and a["fsi_region"] == [0.0, 1.0, 2.0, 3.5] and "config" not in a
    b = parse(["-p", "cylinder", "-c", str(cfg), "-dt", "0.01", "--new-arguments", "mesh_path=other.h5"])
    assert b["dt"] == 0.01 and b["T"] == 0.5 and b["mesh_path"] == "other.h5"
    # the file may name the problem (the command line's -p wins, its absence does not), and a `#` / `;` inside a
    # value - a path - belongs to the value
    cfg2 = tmp_path / "p.config"
    cfg2.write_text("problem = aneurysm\nmesh_path = /data/run#3/mesh;v2.h5  # trailing comment\n; another comment\n")
    c = parse(["-c", str(cfg2)])
    assert c["problem"] == "aneurysm" and c["mesh_path"] == "/data/run#3/mesh;v2.h5"
    assert parse(["-c", str(cfg2), "-p", "avf"])["problem"] == "avf"
    assert parse([])["problem"] == "offset_stenosis"
