"""Benchmark of the hot path: Newton iterations per second of the monolithic ALE-FSI time step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tets T]

One "step" = one time step of the offset-stenosis problem exactly as the product driver runs it
(``vasp_amd.monolithic.advance``: pre_solve data, quasi-Newton solve through the C-ABI, state shift, post_solve
diagnostics; only file output is left out) on a synthetic offset-stenosis mesh of about T tetrahedra
(default 1 M = BASELINE.json configs[1]; the mesh comes from vasp_amd.meshgen because the reference tree has no mesh of
that size).  All inputs are resident in HBM before the timed region; the host<->device traffic inside it is the per-step
Dirichlet values, a few scalars per Krylov iteration and the probe / inlet values post_solve prints.

N > 1: ``python bench.py --gpus N`` starts N ranks itself (``python -m torch.distributed.run``, one per GPU, before
anything touches a GPU) unless it already runs as one of them (RANK / WORLD_SIZE set by the launcher).  The SAME problem
is partitioned by elements across the N ranks (vasp_amd/partition.py): every rank assembles and solves on
the cells of the nodes it owns, one owner->ghost halo exchange and a few scalar all-reduces per Krylov iteration go over
RCCL; total work is fixed, so `scaling` is "strong" and `value` is the Newton rate of the one job.
(VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 rehearses N ranks on a single card with host-staged exchanges.)

Storage precisions.  BASELINE.json configs[1] reads "FP64": the HEADLINE run (``--storage fp64``, the default) keeps every array
of the solver in FP64 (Krylov basis, the Jacobian in every product, Schur sweeps; FP32 only where the preconditioner's sweeps ran
in FP32 from round 1) and reports ``dtype: "f64"``.  The mixed-storage mode the library selects by itself (FP32 Krylov basis and
Jacobian copy inside the iterations, FP16 preconditioner records; Newton residual, Jacobian, every accumulation and every
linear-solve verdict FP64) is the same 20 steps once more in a child process: ``value_mixed_storage`` (``--storage mixed`` makes
it the main run, with ``dtype`` saying so).

``--tets-per-gpu T`` (BASELINE.json configs[4]: 10 M tets on 8 GPUs): the mesh grows with the number of ranks, ``scaling: "weak"``.

Output: ONE JSON line on rank 0: metric, value, `roofline` of the kernel group with the largest share of GPU time in the
timed region (orthogonalisation kernels included, their bytes exact from the number of columns streamed), `kernels` (the
whole table the choice was made from), `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Kernel groups of the `kernels` table -> the kernel names of a rocprofv3 trace that belong to them (a name that ends inside
# its template argument list matches every instantiation).  tools/pmc_traffic.py sums the PMC counters over the same groups, so
# that "HBM traffic per launch" and "algorithmic bytes per launch" of a group are averages over the same set of launches: the
# orthogonalisation group holds the FP32 column kernels AND the FP64-window kernels, the outer product both value types.
KERNEL_GROUPS = {"k_gcr_dots": ["k_gcr_dots<", "k_gcr_axpy<"],
                 "k_gcr_flush": ["k_gcr_flush<"],
                 "k_spmv_node6": ["k_spmv_node6<", "k_spmv_node6p<", "k_spmv_node6c<", "k_spmv_node6pc<", "k_spmv_prow<", "k_spmv<0,"],
                 "k_spmv_tiled_f32<1>": ["k_spmv_tiled_f32<1>"], "k_spmv_tiled_f32<3>": ["k_spmv_tiled_f32<3>"],
                 "k_sweep_tiled_f32<1>": ["k_sweep_tiled_f32<1>"], "k_sweep_tiled_f32<3>": ["k_sweep_tiled_f32<3>"],
                 "k_sweep_tiled_h<1>": ["k_sweep_tiled_h<1>"], "k_sweep_tiled_h<3>": ["k_sweep_tiled_h<3>"], "k_sweep_sb_h": ["k_sweep_sb_h"],
                 "k_sweep_sb_b3<0>": ["k_sweep_sb_b3<0>"], "Schur": ["k_sweep_schur_tiled", "k_sweep_csr_mixed", "k_sweep_csr_f32", "k_spmv<2,"],
                 "k_residual": ["k_residual<", "k_residual_gather"], "k_jacobian": ["k_jacobian<2,"]}
FP64_STORAGE_ENV = {"FSI_KRYLOV_FP32": "0", "FSI_OPERATOR_FP32": "0", "FSI_SCHUR_FP32": "0", "FSI_SWEEPS_FP16": "0"}


def self_launch(args) -> int:
    """--gpus N without a launcher: hand over to torch.distributed.run as a CHILD process (this process has not touched
    a GPU and never will), relay its output and exit code."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--tets", str(args.tets),
           "--dt", str(args.dt), "--storage", args.storage] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []) \
          + (["--tets-per-gpu", str(args.tets_per_gpu)] if args.tets_per_gpu else [])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def other_storage_run(args, storage: str) -> dict:
    """The same workload once more in a fresh child process in the OTHER storage mode ("mixed" beside the FP64 headline, "fp64"
    beside a mixed main run): all arithmetic on the Newton level and every linear-solve verdict is FP64 in both; what differs is
    the storage of the Krylov basis, of the Jacobian copy used inside the iterations and of the preconditioner's matrices.  This
    process has released its context before the child starts."""
    cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--tets", str(args.tets), "--dt", str(args.dt), "--no-cpu-baseline", "--storage", storage, "--no-side-line"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    if storage == "mixed":          # this process exported the all-FP64 switches for its own context: the child must not inherit them
        env = {k: v for k, v in env.items() if k not in FP64_STORAGE_ENV}
    t0 = time.perf_counter()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True)
    line = next((ln for ln in reversed(res.stdout.splitlines()) if ln.startswith("{")), None)
    if res.returncode != 0 or line is None:
        return {"error": f"child exited with {res.returncode}", "stderr_tail": res.stderr[-400:]}
    j = json.loads(line)
    return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "newton_iterations": j["newton_iterations"],
            "krylov_iterations": j["krylov_iterations"], "phase_ms": j["phase_ms"], "dtype": j["dtype"],
            "env": FP64_STORAGE_ENV if storage == "fp64" else {},
            "storage_precisions": j["config"]["storage_precisions"], "roofline": j["roofline"], "wall_s": time.perf_counter() - t0}


def cpu_baseline(desc, gpu: dict, budget_s: float = 25.0):
    """The CPU port of the same algorithm (oracle/cpu_port.py) timed on the host cores, two legs:
    (i) ``kernels``: residual assembly, Jacobian element arithmetic and the monolithic CSR product ON THE BENCH MESH ITSELF
    (``desc``: the 1.12 M-tet mesh the GPU run just used), C under OpenMP on all cores, seconds each, in tets/s and GB/s with the
    GPU kernels' figures of this run beside them; (ii) the whole quasi-Newton loop with the HIP path's field-split preconditioner
    (scipy incomplete LU as inner solves) and recycled GCR on a SMALL mesh of the same generator (``krylov_sample_tets``) - its
    serial inner solves do not fit the bench mesh into a bench run.  `value` is leg (ii)'s Newton rate (the metric's unit)."""
    from oracle.cpu_port import full_mesh_kernels, timed_krylov_run
    sample = int(os.environ.get("VASPFSI_CPU_BASELINE_TETS", 48000))
    out = timed_krylov_run(budget_s, tets=sample)
    out["krylov_sample_tets"] = out.pop("tets")
    out["krylov_sample_dofs"] = out.pop("dofs")
    if not os.environ.get("VASPFSI_NO_CPU_KERNEL_LEG"):
        k = full_mesh_kernels(desc, budget_s=float(os.environ.get("VASPFSI_CPU_KERNEL_BUDGET_S", 15.0)))
        k["gpu_same_run"] = gpu            # the GPU kernels of this run on the same mesh, same units
        out["kernels"] = k
        out["sample"] = (f"(i) kernels: the bench mesh itself, {k['tets']} tets - residual assembly {k['residual_assembly']['assemblies']}x "
                         f"whole mesh, Jacobian element arithmetic on {k['jacobian_elements']['tets']} consecutive tets, "
                         + (f"CSR product {k['spmv']['products']}x on the mesh's own {k['spmv']['nnz']}-entry pattern" if "nnz" in k["spmv"] else "CSR product skipped")
                         + f", {k['cores']} threads; (ii) Newton loop: " + out["sample"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # BASELINE.json configs[1]: 20 time steps (one Jacobian lifetime)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tets", type=int, default=int(os.environ.get("VASPFSI_BENCH_TETS", 1000000)))
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--storage", choices=("fp64", "mixed", "default"), default="fp64",
                    help="fp64 (the headline: BASELINE.json configs[1] says FP64): every storage-precision choice off - FP64 Krylov "
                         "basis, FP64 Jacobian in every product, FP64 Schur sweeps, FP32 instead of FP16 sweep matrices; mixed "
                         "(= default, the library's own policy): FP32 basis and Jacobian copy inside the iterations, FP16 records")
    ap.add_argument("--no-side-line", "--no-fp64-line", dest="no_side_line", action="store_true",
                    help="skip the second run in the other storage mode (value_mixed_storage / value_fp64_storage)")
    ap.add_argument("--tets-per-gpu", type=int, default=0,
                    help="weak scaling (BASELINE.json configs[4]: 10 M tets on 8 GPUs = 1 250 000 per GPU): the mesh has about "
                         "this many tets per rank, `scaling` is \"weak\"")
    ap.add_argument("--profile-host", action="store_true", help="wall time of pre_solve / boundary data / Newton solve / shift / "
                                                                "post_solve per step in the JSON line (host_ms_per_step)")
    args = ap.parse_args()
    if args.storage == "default":
        args.storage = "mixed"
    if args.tets_per_gpu:
        args.tets = args.tets_per_gpu * max(1, args.gpus)
    if args.storage == "fp64":           # read by fsi_create (getenv), so set before the library is touched
        os.environ.update(FP64_STORAGE_ENV)

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between the ranks' devices on this driver stack

    import torch
    from vasp_amd.dist import aggregate, init_from_env
    # VASPFSI_FORCE_PARTITION=1: a single rank still goes through DistBackend and the process group (wire-path check)
    force = bool(os.environ.get("VASPFSI_FORCE_PARTITION"))
    rank, local_rank, world, dist = init_from_env(backend=os.environ.get("VASPFSI_DIST_BACKEND"), force_group=force)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("VASPFSI_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from vasp_amd.capi import HipBackend
    from vasp_amd.meshgen import write_mesh
    from vasp_amd.monolithic import advance, prepare

    partitioned = world > 1 or force
    # N > 1: rank 0 generates the mesh, runs the problem file's hooks and drives the time loop; the other ranks receive their
    # part of the element partition and serve the collective calls rank 0 announces (vasp_amd/partition.py: start_driver /
    # run_worker) - no rank but 0 ever holds the whole mesh.  VASPFSI_SYMMETRIC=1: every rank builds everything (rounds 1-3).
    driver = partitioned and not os.environ.get("VASPFSI_SYMMETRIC")
    if driver and rank != 0:
        from vasp_amd.partition import run_worker
        run_worker(dist, device=local_rank)
        dist.destroy_process_group()
        return

    tmp = Path(tempfile.mkdtemp(prefix=f"vaspfsi_bench_r{rank}_"))
    mesh_path = tmp / "stenosis.h5"
    t_setup = time.perf_counter()
    write_mesh(mesh_path, args.tets, seed=0)
    T_end = args.dt * (args.steps + args.warmup)
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(
            ["-p", "offset_stenosis", "-dt", str(args.dt), "-T", str(T_end), "--theta", "0.501", "--verbose", "False",
             "--folder", str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={mesh_path}"])
    if driver:
        from vasp_amd.partition import start_driver
        hb = start_driver(desc, dist, device=local_rank, lin_max_it=int(os.environ.get("VASPFSI_LIN_MAX_IT", 4000)))
    elif partitioned:
        from vasp_amd.partition import DistBackend
        hb = DistBackend(desc, dist, device=local_rank, lin_max_it=int(os.environ.get("VASPFSI_LIN_MAX_IT", 4000)))
    else:
        hb = HipBackend(desc, device=local_rank)
    for which, fn in ns["dvp_"].items():          # post_solve diagnostics run on the device, as in monolithic.run
        fn.backend, fn.which = hb, which
    ns["backend"] = hb
    ns_cheb = (int(os.environ.get("FSI_CHEB_S", 300)), int(os.environ.get("FSI_CHEB_F", 4)))
    mesh = ns["mesh"]
    setup_s = time.perf_counter() - t_setup
    sink = io.StringIO()

    def one_step():
        with contextlib.redirect_stdout(sink):       # the log lines of pre_solve / post_solve are produced, not shown
            hist = advance(ns, hb, bc_values, pressure, hook, 0, out=lambda *a: None)
        ns["counter"] += 1
        sink.seek(0)
        sink.truncate()
        return hist

    def barrier():
        if driver:
            hb.barrier()                     # the workers are told to meet rank 0 here
            return
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    if args.profile_host:
        ns["_profile"] = {}
    hb.timers(reset=True)
    barrier()
    t0 = time.perf_counter()
    n_newton, n_krylov = 0, 0
    krylov_per_solve = []                    # per time step: Krylov iterations of each Newton iteration
    step_s = []                              # wall time of each timed step (the state shift at its end waits for the device)
    for _ in range(args.steps):
        t_step = time.perf_counter()
        hist = one_step()
        step_s.append(time.perf_counter() - t_step)
        n_newton += len(hist)
        n_krylov += sum(h[3] for h in hist)
        krylov_per_solve.append([int(h[3]) for h in hist])
    barrier()
    elapsed = time.perf_counter() - t0
    tm = hb.timers()
    wire = "cuda" if (dist is None or dist.get_backend() == "nccl") else "cpu"
    if driver:
        elapsed, total_newton = hb.aggregate(elapsed, n_newton)      # max over ranks of the barrier-to-barrier time; rank 0 counted the iterations
        C_all = hb.total_local_cells()                               # cells incl. ghost layers, summed over the ranks
    else:
        elapsed, total_newton = aggregate(dist, elapsed, n_newton, device=wire)
        if partitioned:
            total_newton /= world        # one partitioned job: every rank counted the same Newton iterations
            _, C_all = aggregate(dist, 0.0, len(hb.part.cells), device=wire)       # cells incl. ghost layers, summed

    if rank == 0:
        ndof, nnz = hb.ndof, int(hb.lib.fsi_matrix_nnz(hb.ctx))
        C = mesh.num_cells
        C_rank = len(hb.part.cells) if partitioned else C       # the kernels of this rank run on its local cells / rows
        ndof_rank = hb.part.ndof if partitioned else ndof
        # ---- per kernel group: (GPU time attributed in the timed region [ms], launches, algorithmic bytes per launch) ----
        # (sweep kernels: average of sampled HIP-event launches x launches, the rest: every launch timed)
        solid_fp32, solid_fused = bool(int(tm.get("sweep_flags", 0)) & 4), bool(int(tm.get("sweep_flags", 0)) & 8)
        f_launches = tm["precond_applies"] * ns_cheb[1]                                # fluid velocity sweeps
        sweeps = max(0, tm["inner_vv_iters"] - f_launches)                              # fine-level solid sweeps
        ss_avg = tm["solid_spmv_ms"] / max(1, tm["solid_spmv_calls"])
        db_avg = tm["db_spmv_ms"] / max(1, tm["db_spmv_calls"])
        scalar, tiled = bool(tm["disp_scalar"] & 1), bool(tm["disp_scalar"] & 2)
        db_launches = f_launches + (0 if scalar else tm["inner_dd_iters"])
        sc_avg = tm["sc_spmv_ms"] / max(1, tm["sc_spmv_calls"])
        sc_launches = tm["inner_dd_iters"] if scalar else 0
        sch_avg = tm["schur_ms"] / max(1, tm["schur_calls"])
        sc_bytes = (tm["db_pairs"] * 6.0 + tm["tile_entries"] * 20.0 + tm["db_nodes"] * 27.0) if tiled else \
                   (tm["db_pairs"] * 8.0 + tm["db_nodes"] * 43.0)
        dbf_bytes = (tm["db_pairs"] * 14.0 + tm["tile_entries"] * 20.0 + tm["db_nodes"] * 24.0) if tiled else \
                    (tm["db_pairs"] * 16.0 + tm["db_nodes"] * 40.0)
        flags = int(tm.get("sweep_flags", 0))                            # the context's actual state (FsiTimers.sweep_flags)
        fused = tiled and bool(flags & 1)                                # product + Chebyshev update in one launch (default)
        if fused:                                                        # + d, r, x read, x, r, d' written, product not stored
            sc_bytes += tm["db_nodes"] * 80.0
            dbf_bytes += tm["db_nodes"] * 96.0                            # and the float4 Jacobi scaling
        fp16 = fused and bool(flags & 2)                                 # packed FP16 records: 4 / 8 bytes per pair, 24 per solid block
        if fp16:
            sc_bytes -= tm["db_pairs"] * 2.0
            dbf_bytes -= tm["db_pairs"] * 6.0
        qb = tm["q_elem_bytes"]
        skip_rtol = float((hb.hb if hasattr(hb, "hb") else hb).tuning()["f32_verdict_skip_rtol"])
        # orthogonalisation: every launch of k_gcr_dots / k_gcr_axpy streams m columns of Q (ld * qb bytes each) plus
        # w (read, and written by the update) and r; the exact column count is kept by the library
        q_launches = max(1, tm["ortho_q_launches"])
        q_bytes = (tm["ortho_q_cols"] * tm["ldq"] * qb) / q_launches + ndof_rank * 20.0
        z_launches = max(1, tm["ortho_z_launches"])
        z_bytes = (tm["ortho_z_cols"] * tm["ldz"] * 8.0) / z_launches + ndof_rank * 24.0
        generic = bool(os.environ.get("FSI_SPMV_GENERIC"))
        op32 = int(tm.get("spmv_fp32_calls", 0))
        vbytes = 8.0 - 4.0 * op32 / max(tm["spmv_calls"], 1)                  # mean bytes per matrix value over the products of the region
        # the d rows in pair form (sweep_flags bit 6: adopted by the last refresh's check): their 3 x (6 deg + pdeg) stored values per
        # node are not read; 6 values + one neighbour rank per node pair are
        drows = bool(flags & 64) and not generic
        drows_skipped = 3.0 * (6.0 * tm["node_pairs"] + tm["node_vertex_pairs"]) if drows else 0.0
        kernels = {
            f"k_gcr_dots + k_gcr_axpy (Gram-Schmidt against the kept directions: Q in FP{8 * qb} streamed once per launch, "
            f"mean {tm['ortho_q_cols'] / q_launches:.0f} columns)":
                (tm["ortho_ms"], int(tm["ortho_q_launches"]), q_bytes),
            f"k_gcr_flush (x and the new directions from the direction store, mean {tm['ortho_z_cols'] / z_launches:.0f} columns)":
                (tm["flush_ms"], int(tm["ortho_z_launches"]), z_bytes),
            ("k_sweep_tiled_h<1>" if fp16 else "k_sweep_tiled_f32<1>" if fused else "k_spmv_tiled_f32<1>" if tiled else "k_spmv_sc_f32")
            + f" (displacement block sweeps: one {'FP16' if fp16 else 'FP32'} ratio per node pair"
            + (", neighbour vector entries staged in LDS" if tiled else "") + (", fused with the Chebyshev update" if fused else "")
            + "; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)":
                (sc_avg * sc_launches, int(sc_launches), sc_bytes),
            ("k_sweep_tiled_h<3>" if fp16 else "k_sweep_tiled_f32<3>" if fused else "k_spmv_tiled_f32<3>" if tiled else "k_spmv_db_f32")
            + f" (fluid velocity block sweeps, {'FP16' if fp16 else 'FP32'} component-diagonal node blocks" + (", fused with the Chebyshev update" if fused else "")
            + "; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)":
                (db_avg * db_launches, int(db_launches), dbf_bytes),
            ((("k_sweep_sb_h (solid velocity block, fine level: 3x3 block-CSR with FP16 values in 24-byte records, product fused with the Chebyshev update; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)"
               if fp16 else "k_sweep_sb_b3<0> (solid velocity block, fine level: FP32 3x3 block-CSR product fused with the Chebyshev update; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)")
              if solid_fused else "k_spmv_sb (solid velocity block, FP32 3x3 block-CSR, Chebyshev sweeps; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)")
             if solid_fp32 else "k_spmv<1> (solid velocity block, CSR f64, Chebyshev sweeps; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)"):
                (ss_avg * sweeps, int(sweeps),
                 (tm["solid_nnz"] / 9 * (24.0 if fp16 and solid_fused else 40.0) + tm["solid_rows"] * 8.0 + (tm["solid_rows"] / 3 + 1) * 8.0
                  + (tm["solid_rows"] / 3 * (5 * 16.0 + 48.0) if solid_fused else 0.0)) if solid_fp32
                 else (tm["solid_nnz"] * 12.0 + tm["solid_rows"] * 16.0 + (tm["solid_rows"] + 1) * 8.0)),
            "Schur-complement sweep (explicit two-ring pressure matrix: product + Chebyshev update; avg from HIP-event pairs around single launches of the first 16 applications, which run as one chain)":
                (sch_avg * tm["inner_schur_iters"], int(tm["inner_schur_iters"]),
                 tm["schur_nnz"] * (tm["schur_elem_bytes"] + 4.0) + tm["schur_rows"] * 8.0 * 5),
            (f"k_spmv_node6 (monolithic Jacobian, one i32 column per six entries, {op32} of {int(tm['spmv_calls'])} products on the FP32 "
             "copy of the values, the rest FP64; "
             + ("displacement rows from their pair form: six values per node pair instead of 18 + pressure columns; " if drows else "")
             + "+ k_spmv_prow on the pressure rows)"
             if not generic else "k_spmv<0> (monolithic Jacobian, CSR f64 + i32)"):
                (tm["spmv_ms"], tm["spmv_calls"],
                 (nnz - drows_skipped + 6.0 * tm["node_pairs"] * (1 if drows else 0)) * vbytes + nnz * (4.0 / 6.0 if not generic else 4.0)
                 + (4.0 * tm["node_pairs"] if drows else 0.0) + ndof_rank * 16.0 + (ndof_rank + 1) * 8.0),
            # 1 676 B per tet (the scatter counted once; the element-vector round trip of the reproducible
            # gather form - 2 x 512 B per tet - is overhead, not algorithmic)
            "k_residual + k_residual_gather (element residual; per-dof sum of the element vectors in a fixed order)":
                (tm["residual_ms"], tm["residual_calls"], C_rank * 1676.0),
            # F_nonlinear has no d-equation rows: 34 x 64 entries per tet are produced (the 30 x 64 others live in A_pre)
            "k_jacobian<nonlinear> (element Jacobian of F_nonlinear, rows v and p; one launch per colour of the cell colouring)":
                (tm["jacobian_ms"], tm["jacobian_calls"], C_rank * (652.0 + 34 * 64 * 8.0)),
        }
        table = {}
        for name, (ms, calls, nbytes) in kernels.items():
            ach = nbytes / (ms / max(calls, 1) * 1e-3) / 1e9 if ms > 0 and calls > 0 else 0.0
            table[name] = {"gpu_ms": ms, "launches": int(calls), "avg_launch_ms": ms / max(calls, 1),
                           "algorithmic_bytes_per_launch": nbytes, "achieved_GBps": ach, "frac_of_hbm_peak": ach / HBM_PEAK_GBS,
                           "share_of_timed_region": ms / (1e3 * elapsed)}
        dom = max(table, key=lambda k: table[k]["gpu_ms"])
        d = table[dom]
        # HBM traffic per launch: tools/gpu_pmc_r4.sh runs THIS command with --warmup 0 under rocprofv3 --pmc
        # (FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on known-byte streams) and tools/pmc_traffic.py stores, per
        # kernel group, the PMC bytes summed over every launch of the group next to the algorithmic bytes the profiled run's own
        # JSON line reports for the same launches (with no warm-up the library's counters cover the whole process, so the two
        # launch sets are identical).  Their ratio is what the counters say about wasted traffic; `traffic` below is that ratio
        # applied to this run's algorithmic bytes per launch, i.e. on the basis of `algorithmic_bytes_per_launch`.
        traffic, traffic_source, traffic_ratio = None, None, None
        # (a profile of the same storage mode only: r05_pmc_traffic_<mode>.json; round 4's file was a mixed-storage run)
        cands = [ROOT / "profiles" / f"r05_pmc_traffic_{args.storage}.json"] + ([ROOT / "profiles" / "r04_pmc_traffic.json"] if args.storage == "mixed" else [])
        pmc = next((q for q in cands if q.exists()), None)
        if pmc is not None and world == 1:
            with contextlib.suppress(Exception):
                pj = json.loads(pmc.read_text())
                if pj.get("tets") == C and pj.get("dofs") == ndof:
                    for key, g in pj.get("groups", {}).items():
                        if dom.startswith(key) and g.get("algorithmic_bytes_per_launch"):
                            traffic_ratio = g["pmc_bytes_per_launch"] / g["algorithmic_bytes_per_launch"]
                            traffic = traffic_ratio * d["algorithmic_bytes_per_launch"]
                            traffic_source = f"profiles/{pmc.name}: group {key}, {g['launches']} launches of the profiled run"
        # the north star's own aggregate: assembly (residual + Jacobian) and the monolithic SpMV together
        agg = [v for k, v in table.items() if k.startswith(("k_residual", "k_jacobian", "k_spmv_node6", "k_spmv<0>"))]
        agg_bytes = sum(v["algorithmic_bytes_per_launch"] * v["launches"] for v in agg)
        agg_ms = sum(v["gpu_ms"] for v in agg)
        assembly_spmv = {"kernels": "k_residual + k_residual_gather, k_jacobian<nonlinear>, monolithic SpMV (north_star: assembly + SpMV >= 40 % of the HBM roofline)",
                         "algorithmic_bytes": agg_bytes, "gpu_ms": agg_ms, "achieved_GBps": agg_bytes / (agg_ms * 1e-3) / 1e9 if agg_ms > 0 else 0.0,
                         "peak_GBps": HBM_PEAK_GBS, "frac": (agg_bytes / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if agg_ms > 0 else 0.0,
                         "per_kernel_frac": {k.split(" ")[0]: v["frac_of_hbm_peak"] for k, v in table.items()
                                             if k.startswith(("k_residual", "k_jacobian", "k_spmv"))}}
        out = {
            "metric": "Newton-iterations/sec (offset_stenosis, monolithic ALE-FSI step)",
            "value": total_newton / elapsed, "unit": "Newton-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            # one problem partitioned over the N ranks: of fixed size ("strong": the north star's >= 5x at 8 GPUs), or - with
            # --tets-per-gpu - of a size that grows with N ("weak": BASELINE.json configs[4])
            "scaling": "weak" if args.tets_per_gpu else "strong",
            "vs_baseline": None,
            # what ran: "f64" only when every array of the solver is stored in FP64 (the headline); the mixed mode computes its
            # Newton residual, Jacobian, accumulations and verdicts in FP64 on FP32 / FP16 STORAGE of the Krylov basis, the Jacobian
            # copy of the inner products and the preconditioner's records, and says so
            "dtype": "f64" if args.storage == "fp64" else "f64 (Newton residual, Jacobian, accumulations, verdicts) / f32 Krylov-basis and Jacobian-copy storage / f16 preconditioner records",
            "data": "synthetic",
            "config": {"workload": f"offset_stenosis synthetic mesh, {C} tets, {ndof} dofs, dt={args.dt}, theta=0.501, "
                                   f"quasi-Newton atol=rtol=1e-6 recompute_tstep={ns['recompute_tstep']}, "
                                   f"per step: pre_solve, Newton solve, shift, post_solve (no file output); storage: "
                                   + ("all-FP64 (BASELINE.json configs[1])" if args.storage == "fp64" else "mixed (FP32 Krylov basis / Jacobian copy, FP16 preconditioner records)")
                                   + (f"; weak scaling, {args.tets_per_gpu} tets per GPU asked for" if args.tets_per_gpu else ""),
                       "tets": C, "dofs": ndof, "matrix_nnz": nnz, "parallelism": (f"element partition over {world} ranks (node slabs, ghost-layer cells {C_all / C - 1:.1%}), "
                                       f"halo + all-reduce over {dist.get_backend()}") if partitioned else "1 GPU",
                       "rank0_matrix_nnz": nnz,
                       "assembly": (f"bitwise reproducible: residual summed per dof over its cells in a fixed order, Jacobian in "
                                    f"{int(tm['assembly_colours'])} launches (cell colours)" if tm["assembly_colours"] > 0
                                    else "unordered atomics (FSI_ASSEMBLY=atomic)"),
                       "krylov_storage": f"Q FP{8 * qb}, directions FP64, capacity {tm['krylov_cap']}",
                       "storage_precisions": ("state, residual, Jacobian and every accumulation FP64; Krylov basis "
                                              f"FP{8 * qb}" + (" with an FP64 window of 32 columns" if qb == 4 else "")
                                              + f"; {op32} of {int(tm['spmv_calls'])} outer products on an FP32 copy of the Jacobian values, "
                                              + (f"every linear answer asked for below {skip_rtol:g} judged on the FP64 residual of the FP64 matrix, looser ones "
                                                 "too unless the last verified cycle had recurrence and truth within 1 % of the request "
                                                 f"({int(tm['verdicts_skipped'])} of {int(tm['krylov_solves'])} solves) - Newton's FP64 residual follows; " if qb == 4 else
                                                 "iterations on the FP64 matrix, answers re-judged on b - A x when asked below 1e-8 or when the "
                                                 "kept pairs are at risk (full store, long or stalled cycle, earlier mismatch); ") + "preconditioner "
                                              "sweeps: " + ("FP16" if fp16 else "FP32") + " matrix values, FP32 vectors (Schur: "
                                              + {0: "FP16", 4: "FP32", 8: "FP64"}[int(tm["schur_elem_bytes"])] + " values, FP64 vectors)")},
            "dof_updates_per_s": total_newton * ndof / elapsed,
            "newton_iterations": n_newton, "krylov_iterations": n_krylov, "krylov_per_solve": krylov_per_solve,
            "phase_ms": {k: tm[k] for k in ("residual_ms", "jacobian_ms", "factor_ms", "spmv_ms", "precond_ms", "ortho_ms", "flush_ms", "krylov_ms")},
            "phase_calls": {k: tm[k] for k in ("residual_calls", "jacobian_calls", "factor_calls", "spmv_calls", "precond_calls",
                                               "krylov_solves", "krylov_iters", "inner_vv_iters", "inner_schur_iters", "inner_dd_iters",
                                               "ortho_q_launches", "ortho_q_cols", "ortho_z_launches", "ortho_z_cols")},
            "solver_events": {k: int(tm[k]) for k in ("gcr_arnoldi_steps", "gcr_restarts", "newton_retries", "fp32_fallbacks", "verdicts_skipped",
                                                        "reorth_forced")},
            "run_totals": (hb.hb if hasattr(hb, "hb") else hb).solver_events(),      # since fsi_create (warm-up included): adaptive solves, exact coarse solves ...
            "setup_s": setup_s,
            "host_ms_per_step": ({k: 1e3 * v / args.steps for k, v in ns["_profile"].items()} if args.profile_host else None),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": d["frac_of_hbm_peak"], "traffic": traffic, "traffic_over_algorithmic": traffic_ratio,
                         "traffic_source": traffic_source if traffic is not None else None,
                         "launches": d["launches"],
                         "avg_launch_ms": d["avg_launch_ms"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                         "share_of_timed_region": d["share_of_timed_region"]},
            "assembly_spmv": assembly_spmv,
            "kernels": table,
        }
        if args.steps > 20:          # long runs: the time per step grows with the ramp (more Newton iterations per step); per Jacobian lifetime
            life = int(ns["recompute_tstep"])
            out["per_lifetime"] = [{"steps": f"{a}-{min(a + life, args.steps) - 1}", "ms_per_step": 1e3 * sum(step_s[a:a + life]) / len(step_s[a:a + life]),
                                    "newton_iterations": sum(len(k) for k in krylov_per_solve[a:a + life]),
                                    "krylov_iterations": sum(sum(k) for k in krylov_per_solve[a:a + life])}
                                   for a in range(0, args.steps, life)]
    hb.close()                 # the context's HBM (Krylov store, matrices) is free again before anything else runs
    if rank == 0:
        out["config"]["storage"] = args.storage
        if world == 1 and not args.no_side_line:      # the same steps once more in the other storage mode
            other = "mixed" if args.storage == "fp64" else "fp64"
            side = other_storage_run(args, other)
            out[f"value_{other}_storage"] = side.get("value")        # Newton-iterations/s of the same 20 steps
            out[f"ms_per_step_{other}_storage"] = side.get("ms_per_step")
            out[f"{other}_storage"] = side
        if not args.no_cpu_baseline and world == 1:
            res_k = next(v for k, v in table.items() if k.startswith("k_residual"))
            jac_k = next(v for k, v in table.items() if k.startswith("k_jacobian"))
            spmv_k = next(v for k, v in table.items() if k.startswith("k_spmv"))
            gpu = {"residual_assembly": {"tets_per_s": C / (res_k["avg_launch_ms"] * 1e-3) if res_k["launches"] else None,
                                         "GBps_algorithmic": res_k["achieved_GBps"], "seconds_per_assembly": res_k["avg_launch_ms"] * 1e-3},
                   "jacobian_assembly": {"tets_per_s": C / (jac_k["avg_launch_ms"] * 1e-3) if jac_k["launches"] else None,
                                         "seconds_per_assembly": jac_k["avg_launch_ms"] * 1e-3,
                                         "note": "k_jacobian incl. the scatter into the CSR matrix and the matrix finish"},
                   "spmv": {"GBps_algorithmic": spmv_k["achieved_GBps"], "seconds_per_product": spmv_k["avg_launch_ms"] * 1e-3,
                            "note": "k_spmv_node6: one i32 column per six entries (8.67 B per entry against the CPU port's 12)"
                                    + ("; displacement rows in pair form (a third of the node rows' values not read)" if drows else "")}}
            out["cpu_baseline"] = cpu_baseline(desc, gpu)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
