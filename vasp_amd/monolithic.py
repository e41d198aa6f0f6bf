"""Time-step driver behind the problem-file API: the ``turtleFSI -p <problem>`` replacement.

Host-side counterpart of turtleFSI's ``monolithic.py`` + ``utils/argpar.py`` as VaSP drives them
[REF docs/simulation.md:9-31; tests/test_simulations.py:22-24; SURVEY.md §3.1]: parse the same CLI,
run the seven problem hooks in the same order with the whole namespace as keyword arguments, keep the
same loop (``while t <= T + dt/10``), the same log lines [REF src/vasp/postprocessing/log_plotter.py:71-82]
and the same output tree.  The per-step Newton solve itself is not here: it is one call through the
C-ABI (``include/vaspfsi.h``) into the HIP time-step kernel.
"""
from __future__ import annotations

import argparse
import ast
import importlib
import importlib.util
import json
import re
import sys
import time as _time
from pathlib import Path
from pprint import pprint
from typing import Callable, Dict, List, Optional

import numpy as np

from . import problems as _defaults
from .fem import (DirichletBC, FormTerms, MixedFunction, MixedSpace, RobinTerm, SurfacePressureTerm,
                  resolve_bcs)
from .mesh import FsiMesh
from .output import VisualizationWriter, checkpoint, read_checkpoint

MATERIAL_IDS = {"StVenantKirchoff": 0, "MooneyRivlin": 1}


# ------------------------------------------------------------------------------------------------
# CLI (turtleFSI/utils/argpar.py as used by VaSP)
# ------------------------------------------------------------------------------------------------

def _coerce(text: str):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def parse(argv: Optional[List[str]] = None) -> Dict[str, object]:
    ap = argparse.ArgumentParser(prog="vaspfsi", description="MI355X-native monolithic ALE-FSI solver "
                                 "behind the turtleFSI problem-file API")
    ap.add_argument("-p", "--problem", default=None)       # "offset_stenosis" unless the command line or the -c file names one
    ap.add_argument("-dt", "--time-step", dest="dt", type=float, default=None)
    ap.add_argument("-T", "--end-time", dest="T", type=float, default=None)
    ap.add_argument("-t", "--theta", dest="theta", type=float, default=None)
    ap.add_argument("--atol", type=float, default=None)
    ap.add_argument("--rtol", type=float, default=None)
    ap.add_argument("--max-it", dest="max_it", type=int, default=None)
    ap.add_argument("--lmbda", type=float, default=None)
    ap.add_argument("--recompute", type=int, default=None)
    ap.add_argument("--recompute-tstep", dest="recompute_tstep", type=int, default=None)
    ap.add_argument("--verbose", type=_coerce, default=None)
    ap.add_argument("--folder", default=None)
    ap.add_argument("--sub-folder", dest="sub_folder", default=None)
    ap.add_argument("--restart-folder", dest="restart_folder", default=None)
    ap.add_argument("--save-step", dest="save_step", type=int, default=None)
    ap.add_argument("--save-deg", dest="save_deg", type=int, default=None)
    ap.add_argument("--checkpoint-step", dest="checkpoint_step", type=int, default=None)
    ap.add_argument("--killtime", type=float, default=None)
    ap.add_argument("--new-arguments", dest="new_arguments", nargs="*", default=[])
    ap.add_argument("-c", "--config", dest="config", default=None,
                    help="config file with `key = value` lines (keys: the option names without dashes, or any problem-file "
                         "parameter); the command line wins over the file [REF docs/simulation.md:19-31]")
    ns = ap.parse_args(argv)
    out = {}
    if ns.config is not None:          # turtleFSI's ConfigArgParse behaviour: file < command line
        names = {o.lstrip("-"): a.dest for a in ap._actions for o in a.option_strings}
        types = {a.dest: a.type for a in ap._actions}
        for raw in Path(ns.config).read_text().splitlines():
            # comments: a line that starts with # or ;, or the rest of a line after whitespace + # / ; (a `#` or `;` inside a
            # value - a path, say - belongs to the value, as in ConfigArgParse)
            line = re.sub(r"(^|\s)[#;].*$", "", raw).strip()
            if not line or line.startswith("["):
                continue
            for sep in ("=", ":", None):
                parts = line.split(sep, 1)
                if len(parts) == 2:
                    break
            if len(parts) != 2:
                raise SystemExit(f"{ns.config}: cannot parse {raw!r} (expected `key = value`)")
            key, val = parts[0].strip().lstrip("-"), parts[1].strip()
            dest = names.get(key, key.replace("-", "_"))
            if dest in ("config", "new_arguments"):
                continue
            conv = types.get(dest)
            out[dest] = conv(val) if conv not in (None, _coerce) else _coerce(val)
    out.update({k: v for k, v in vars(ns).items() if v is not None and k not in ("new_arguments", "config")})
    for kv in ns.new_arguments:
        if "=" not in kv:
            raise SystemExit(f"--new-arguments expects key=value, got {kv!r}")
        k, v = kv.split("=", 1)
        out[k] = _coerce(v)
    out.setdefault("problem", "offset_stenosis")
    return out


def load_problem(name: str):
    """Problem lookup, current directory first (as the reference's ``exec("from <problem> import *")``)."""
    local = Path.cwd() / (name if name.endswith(".py") else name + ".py")
    if local.exists():
        spec = importlib.util.spec_from_file_location(local.stem, local)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    return importlib.import_module(f"vasp_amd.problems.{name}")


# ------------------------------------------------------------------------------------------------
# description handed to the time-step kernel
# ------------------------------------------------------------------------------------------------

def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def build_properties(v: dict):
    """``fluid_properties`` / ``solid_properties`` lists from scalar-or-list parameters."""
    dx_f = _as_list(v["dx_f_id"])
    if not v.get("fluid_properties"):
        rho, mu = _as_list(v["rho_f"]), _as_list(v["mu_f"])
        v["fluid_properties"] = [dict(dx_f_id=dx_f[i], rho_f=rho[i if len(rho) > 1 else 0],
                                      mu_f=mu[i if len(mu) > 1 else 0]) for i in range(len(dx_f))]
    elif isinstance(v["fluid_properties"], dict):
        v["fluid_properties"] = [v["fluid_properties"]]
    dx_s = _as_list(v["dx_s_id"])
    if not v.get("solid_properties"):
        pick = lambda key, i: _as_list(v[key])[i if len(_as_list(v[key])) > 1 else 0]
        v["solid_properties"] = [dict(dx_s_id=dx_s[i], material_model=pick("material_model", i),
                                      rho_s=pick("rho_s", i), mu_s=pick("mu_s", i), lambda_s=pick("lambda_s", i))
                                 for i in range(len(dx_s))]
    elif isinstance(v["solid_properties"], dict):
        v["solid_properties"] = [v["solid_properties"]]
    return v["fluid_properties"], v["solid_properties"]


def build_description(mesh: FsiMesh, v: dict, bcs, F_solid_linear) -> dict:
    """Plain-array description of the discrete problem (what ``fsi_create`` receives)."""
    fluid_properties, solid_properties = v["fluid_properties"], v["solid_properties"]
    kind = -np.ones(mesh.num_cells, dtype=np.int32)
    region = np.zeros(mesh.num_cells, dtype=np.int32)
    for r, fp in enumerate(fluid_properties):
        sel = mesh.cell_markers == fp["dx_f_id"]
        kind[sel], region[sel] = 0, r
    for r, spp in enumerate(solid_properties):
        sel = mesh.cell_markers == spp["dx_s_id"]
        kind[sel], region[sel] = 1, r
    if (kind < 0).any():
        bad = np.unique(mesh.cell_markers[kind < 0])
        raise ValueError(f"cells with domain markers {bad.tolist()} belong to neither dx_f_id nor dx_s_id")
    solid_rows = []
    for spp in solid_properties:
        model = spp.get("material_model", "StVenantKirchoff")
        if model not in MATERIAL_IDS:
            raise NotImplementedError(f"material_model {model!r}")
        solid_rows.append((float(spp["rho_s"]), float(spp.get("mu_s", 0.0)), float(spp.get("lambda_s", 0.0)),
                           float(spp.get("C10", 0.0)), float(spp.get("C01", 0.0)), float(spp.get("C11", 0.0))))
    desc = dict(
        coords=mesh.coords, tets=mesh.tets, tet_nodes=mesh.tet_nodes, num_nodes=mesh.num_nodes,
        cell_kind=kind, cell_region=region,
        fluid_props=[(float(fp["rho_f"]), float(fp["mu_f"])) for fp in fluid_properties],
        solid_props=solid_rows,
        solid_models=[MATERIAL_IDS[spp.get("material_model", "StVenantKirchoff")] for spp in solid_properties],
        dt=float(v["dt"]), theta=float(v["theta"]),
    )
    pterms = [tm for tm in F_solid_linear if isinstance(tm, SurfacePressureTerm)]
    if any(tm.pressure is not pterms[0].pressure for tm in pterms):
        raise NotImplementedError("surface-pressure terms with different pressure expressions")
    if pterms:                                      # several dS(id) terms of one pressure (avf: both fsi ids) -> one facet list
        parts = [tm.facets(mesh) for tm in pterms]
        fids = np.concatenate([f for f, _ in parts])
        plus = np.concatenate([c for _, c in parts])
        desc["pressure_facets"] = mesh.facet_nodes[fids]
        desc["pressure_facet_cell"] = plus
    rterms = [tm for tm in F_solid_linear if isinstance(tm, RobinTerm)]
    if rterms:
        rf, rk, rc = [], [], []
        for tm in rterms:
            fids = np.nonzero(np.asarray(tm.boundaries) == tm.marker)[0]
            rf.append(mesh.facet_nodes[fids])
            rk.append(np.full(len(fids), tm.k_s))
            rc.append(np.full(len(fids), tm.c_s))
        desc["robin_facets"] = np.concatenate(rf)
        desc["robin_k"] = np.concatenate(rk)
        desc["robin_c"] = np.concatenate(rc)
    bc_dofs, bc_values = resolve_bcs(bcs, mesh.num_dofs)
    desc["bc_dofs"] = bc_dofs
    return desc, bc_values, (pterms[0].pressure if pterms else None)


# ------------------------------------------------------------------------------------------------
# the driver
# ------------------------------------------------------------------------------------------------

def default_backend(desc):
    """The HIP time-step kernel: one context on one GPU, or - when the process was started as one of N ranks
    (``torchrun`` / ``python -m torch.distributed.run``: WORLD_SIZE > 1, the counterpart of the reference's
    ``mpirun -np N turtleFSI ...`` [REF docs/simulation.md:16,30]) - this rank's part of the element partition."""
    import os
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch
        from .dist import init_from_env
        from .partition import DistBackend, start_driver
        rank, local_rank, world, dist = init_from_env(backend=os.environ.get("VASPFSI_DIST_BACKEND"))
        if os.environ.get("VASPFSI_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if os.environ.get("VASPFSI_SYMMETRIC"):          # rounds 1-3: every rank read the mesh and ran the hooks
            return DistBackend(desc, dist, device=local_rank)
        return start_driver(desc, dist, device=local_rank)       # this is rank 0: the other ranks are in run_worker (see run)
    from .capi import HipBackend   # raises loudly if libvaspfsi.so or the GPU is missing
    return HipBackend(desc)


def prepare(argv: Optional[List[str]] = None):
    """Everything ``run`` does before the time loop: parameters, folders, mesh, hooks up to ``create_bcs``.

    Returns (ns, desc, bc_values, pressure, hook): the namespace, the plain-array problem description the kernel
    library consumes, a callable giving the current Dirichlet values, the interface-pressure object (or None) and the
    hook lookup.
    """
    args = parse(argv)
    problem = load_problem(args.pop("problem"))
    hook = lambda name: getattr(problem, name, getattr(_defaults, name))

    v = {k: (list(x) if isinstance(x, list) else x) for k, x in _defaults.default_variables.items()}
    v = hook("set_problem_parameters")(default_variables=v, **args)
    v.update(args)
    build_properties(v)
    ns: Dict[str, object] = dict(v)
    ns["default_variables"] = v
    if ns["verbose"]:
        pprint(v)

    # folders ---------------------------------------------------------------------------------
    folder = Path(str(ns["folder"]))
    if ns.get("restart_folder"):
        # a restarted run continues in the folder it restarts from: the checkpoint is overwritten and the visualization
        # series goes on in ONE new file per field, <name>_run_N.h5, behind the same xdmf (what output_file_lists expects of a restarted series
        # [REF src/vasp/postprocessing/postprocessing_common.py:63-121])
        results = Path(str(ns["restart_folder"]))
    elif ns.get("sub_folder") is not None:
        results = folder / str(ns["sub_folder"])
    else:
        existing = [int(p.name) for p in folder.glob("*") if p.name.isdigit()] if folder.exists() else []
        results = folder / str(max(existing) + 1 if existing else 1)
    for sub in ("Checkpoint", "Mesh", "Visualization"):
        (results / sub).mkdir(parents=True, exist_ok=True)
    ns.update(results_folder=results, visualization_folder=results / "Visualization",
              checkpoint_folder=results / "Checkpoint")

    # mesh + space -------------------------------------------------------------------------------
    mesh, domains, boundaries = hook("get_mesh_domain_and_boundaries")(**ns)
    mesh.cell_markers, mesh.facet_markers = domains, boundaries
    mesh.write(results / "Mesh" / "mesh.h5")
    DVP = MixedSpace(mesh)
    n_dof = mesh.num_dofs
    state = {k: np.zeros(n_dof) for k in ("n", "n-1")}
    dvp_ = {k: MixedFunction(mesh, x, which=k) for k, x in state.items()}
    ns.update(mesh=mesh, domains=domains, boundaries=boundaries, DVP=DVP, dvp_=dvp_, psi="psi", phi="phi",
              gamma="gamma", F_solid_linear=FormTerms(), F_fluid_linear=FormTerms(),
              t=float(ns["t"]), counter=int(ns["counter"]), _state=state)
    if ns.get("robin_bc"):
        ds_ids, k_s, c_s = _as_list(ns["ds_s_id"]), _as_list(ns["k_s"]), _as_list(ns["c_s"])
        for i, marker in enumerate(ds_ids):      # scalar k_s / c_s apply to every listed surface [REF avf.py:82-84]
            ns["F_solid_linear"] += RobinTerm(boundaries, marker, k_s[i if len(k_s) > 1 else 0], c_s[i if len(c_s) > 1 else 0])

    upd = hook("initiate")(**ns)
    ns.update(upd or {})
    upd = hook("create_bcs")(**ns)
    ns.update(upd or {})

    desc, bc_values, pressure = build_description(mesh, v, ns["bcs"], ns["F_solid_linear"])
    return ns, desc, bc_values, pressure, hook


def advance(ns, backend, bc_values, pressure, hook, first_step_num: int, out=print) -> list:
    """One time step of the reference's loop body: ``t += dt``; ``pre_solve``; Dirichlet data and
    interface pressure to the device; the quasi-Newton solve behind the C-ABI; state shift; ``post_solve``.  File output
    and the step counter stay with the caller.  ``bench.py`` times exactly this function."""
    prof = ns.get("_profile")             # bench.py --profile-host: wall time of the five parts of a step
    tick = _time.perf_counter
    t0 = tick()
    ns["t"] = ns["t"] + float(ns["dt"])
    upd = hook("pre_solve")(**ns)
    ns.update(upd or {})
    t1 = tick()
    backend.set_dirichlet_values(bc_values())
    backend.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
    t2 = tick()
    hist = backend.newton_solve(counter=ns["counter"], first_step_num=first_step_num,
                                log=out if ns["verbose"] else None,
                                **{k: ns[k] for k in ("atol", "rtol", "max_it", "lmbda", "recompute", "recompute_tstep")})
    t3 = tick()
    backend.shift()                       # dvp_["n-1"] <- dvp_["n"]
    for fn in ns["dvp_"].values():        # the host copies are refreshed only if somebody reads them
        fn.mark_stale()
    t4 = tick()
    upd = hook("post_solve")(**ns)
    ns.update(upd or {})
    if prof is not None:
        for key, dt_ in (("pre_solve", t1 - t0), ("boundary_data", t2 - t1), ("newton_solve", t3 - t2), ("shift", t4 - t3),
                         ("post_solve", tick() - t4)):
            prof[key] = prof.get(key, 0.0) + dt_
    return hist


def stop_controls(results: Path, killtime, t_loop: float, agree=None, rank0: bool = True, out=print,
                  poll_s: float = 5.0) -> bool:
    """turtleFSI's stop controls after a time step: the wall-clock budget ``killtime`` and the sentinel files a user drops
    into the results folder (``killturtle``: checkpoint and stop; ``pauseturtle``: wait until it is removed).  Returns
    True when the run has to write a checkpoint and stop.

    Every rank looks at its own clock and at the shared folder; the job acts on the OR of the local flags (``agree``:
    ``DistBackend.agree_flags``), so all ranks leave the loop - or enter the collective gather of the stop checkpoint - in
    the same step.  turtleFSI sums the flag over MPI before acting for the same reason.  Rank 0 removes the sentinel only
    after that agreement."""
    import contextlib
    agree = agree or (lambda flags: [bool(f) for f in flags])
    late = killtime is not None and _time.perf_counter() - t_loop > float(killtime)
    kill, late = agree([(results / "killturtle").exists(), late])
    stop = False
    if late:
        out("Reached killtime = %s s: writing a checkpoint and stopping" % killtime)
        stop = True
    if kill:
        out("killturtle found: writing a checkpoint and stopping")
        if rank0:
            with contextlib.suppress(OSError):
                (results / "killturtle").unlink()
        stop = True
    while agree([(results / "pauseturtle").exists()])[0]:
        _time.sleep(poll_s)
    return stop


SOLVER_EVENT_KEYS = ("newton_retries", "fp32_fallbacks", "gcr_restarts")


def solver_events(backend) -> Dict[str, int]:
    """What the linear solver had to do beyond iterating, as the library counts it (FsiTimers): Newton iterations whose
    solve failed on a stale Jacobian and succeeded after a refresh the reference's policy would not have made
    (`newton_retries` - such an iteration is logged as "Compute Jacobian matrix"), Jacobian lifetimes that lost the FP32
    Krylov basis (`fp32_fallbacks`), solves that dropped the recycled directions (`gcr_restarts`).  Empty for a backend
    without timers (the oracle backend of the CPU tests)."""
    events = getattr(backend, "solver_events", None)
    if events is None:
        return {}
    ev = events()              # fsi_get_solver_events: run totals, no timer is resolved and a timer reset does not zero them
    return {k: int(ev[k]) for k in SOLVER_EVENT_KEYS if k in ev}


def _rank() -> int:
    import os
    return int(os.environ.get("RANK", 0))


def run(argv: Optional[List[str]] = None, backend_factory: Callable = default_backend, out=print):
    """Run one simulation; returns the final namespace (for tests).  ``out`` receives solver log lines."""
    import builtins
    import contextlib
    import io
    import os
    rank0 = _rank() == 0
    world = int(os.environ.get("WORLD_SIZE", 1))
    driver_mode = world > 1 and backend_factory is default_backend and not os.environ.get("VASPFSI_SYMMETRIC")
    if driver_mode and not rank0:
        # Ranks 1 .. N-1 of `python -m torch.distributed.run --nproc-per-node N -m vasp_amd.monolithic ...`: they never read the
        # mesh, build no function space and run no hook; rank 0 sends each its part of the element partition (as DOLFIN
        # distributes a mesh it read once [REF src/vasp/simulations/offset_stenosis.py:20-23]) and announces every collective
        # call of the time loop (vasp_amd/partition.py: start_driver / run_worker).
        import torch
        from .dist import init_from_env
        from .partition import run_worker
        _, local_rank, _, dist = init_from_env(backend=os.environ.get("VASPFSI_DIST_BACKEND"))
        if os.environ.get("VASPFSI_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        run_worker(dist, device=local_rank)
        return {"worker": True}
    if not rank0:                                 # the reference guards its prints with MPI.rank == 0
        out = lambda *a, **k: None
    quiet = contextlib.nullcontext() if rank0 else contextlib.redirect_stdout(io.StringIO())
    with quiet:
        ns, desc, bc_values, pressure, hook = prepare(argv)
    state = ns["_state"]
    backend = backend_factory(desc)
    ns["backend"] = backend
    if driver_mode:
        # whatever ends this function - the last time step or an exception in a hook or in the solver - the workers' serve
        # loops must end with it
        try:
            return _time_loop(ns, backend, bc_values, pressure, hook, out, rank0, quiet)
        finally:
            with contextlib.suppress(Exception):
                backend.close()
    return _time_loop(ns, backend, bc_values, pressure, hook, out, rank0, quiet)


def _time_loop(ns, backend, bc_values, pressure, hook, out, rank0, quiet):
    """The part of ``run`` behind the backend's construction: restart, the time loop, output, ``finished``."""
    state = ns["_state"]
    mesh = ns["mesh"]
    for which, fn in ns["dvp_"].items():          # per-step diagnostics of post_solve run on the device
        fn.backend, fn.which = backend, which
    restart_run = 0
    if ns.get("restart_folder"):                  # --restart-folder: resume from Checkpoint/ of an earlier run
        ck = Path(str(ns["restart_folder"])) / "Checkpoint"
        meta = json.loads((ck / "default_variables.json").read_text())
        state["n"][:] = read_checkpoint(ck, mesh)
        state["n-1"][:] = state["n"]
        backend.set_state("n", state["n"])
        backend.set_state("n-1", state["n-1"])
        ns["t"], ns["counter"] = float(meta["t"]), int(meta["counter"])
        runs = [int(p.stem.rsplit("_", 1)[1]) for p in Path(ns["visualization_folder"]).glob("velocity_run_*.h5")]
        restart_run = 1 + max(runs, default=0)
    viz = None
    if ns.get("save_step") and rank0:
        viz = VisualizationWriter(ns["visualization_folder"], mesh, ns["save_deg"], run_index=restart_run)
    first_step_num = ns["counter"]

    dt, T = float(ns["dt"]), float(ns["T"])
    killtime = ns.get("killtime")
    results = Path(ns["results_folder"])
    total_newton = 0
    stop = False
    events_seen: Dict[str, int] = {}
    t_loop = _time.perf_counter()
    while ns["t"] <= T + dt / 10 and not stop:
        t0 = _time.perf_counter()
        with quiet:
            hist = advance(ns, backend, bc_values, pressure, hook, first_step_num, out)
        t = ns["t"]
        total_newton += len(hist)
        # turtleFSI's stop controls: wall-clock budget, and the sentinel files a user drops into the results folder
        # (``killturtle``: checkpoint and stop; ``pauseturtle``: wait until it is removed)
        stop = stop_controls(results, killtime, t_loop, getattr(backend, "agree_flags", None), rank0, out)
        if ns.get("checkpoint_step") and (ns["counter"] % int(ns["checkpoint_step"]) == 0 or stop):
            x = ns["dvp_"]["n"].vector()
            if rank0:
                checkpoint(ns["checkpoint_folder"], mesh, x, ns["default_variables"], t, ns["counter"])
        if viz is not None and ns["counter"] % int(ns["save_step"]) == 0:
            viz.write(ns["dvp_"]["n"].vector(), t)
        elif ns.get("save_step") and ns["counter"] % int(ns["save_step"]) == 0:
            ns["dvp_"]["n"].vector()              # partitioned: every rank takes part in the gather
        ns["counter"] += 1
        ev = solver_events(backend)          # the silent refresh-and-retry is visible in the product log
        if any(ev.get(k, 0) > events_seen.get(k, 0) for k in ev):
            out("Linear solver events so far: " + ", ".join(f"{k} = {v}" for k, v in ev.items()))
        events_seen = ev
        out("Solved for timestep %d, t = %.4f in %.1f s" % (ns["counter"], t, _time.perf_counter() - t0))
    if viz is not None:
        viz.close()
    ns["time_loop_seconds"] = _time.perf_counter() - t_loop
    ns["newton_iterations"] = total_newton
    ns["solver_events"] = events_seen
    with quiet:
        hook("finished")(**ns)
    return ns


def main(argv: Optional[List[str]] = None) -> int:
    run(argv)
    return 0


if __name__ == "__main__":
    sys.exit(main())
