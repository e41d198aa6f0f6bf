"""Benchmark of the hot path: Newton iterations per second of the monolithic ALE-FSI time step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tets T]

One "step" = one time step of the offset-stenosis problem (pre_solve data, quasi-Newton solve through the C-ABI, state
shift) on a synthetic offset-stenosis mesh of about T tetrahedra (default 1 M = BASELINE.json configs[1]; the mesh comes
from vasp_amd.meshgen because the reference tree has no mesh of that size).  All inputs are resident in HBM before the
timed region; the only host<->device traffic inside it is the per-step Dirichlet values and a few scalars.

N > 1: the SAME problem is partitioned by elements across the N ranks (vasp_amd/partition.py): every
rank assembles and solves on the cells of the nodes it owns, one owner->ghost halo exchange and a few scalar all-reduces
per Krylov iteration go over RCCL; total work is fixed, so `scaling` is "strong" and `value` is the Newton rate of the
one job.  (VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 rehearses N ranks on a single card with host-staged exchanges.)

Output: ONE JSON line on rank 0 (metric, value, roofline of the dominant kernel, cpu_baseline of the oracle).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(budget_s: float = 25.0):
    """The oracle (numpy restatement, exact sparse LU) timed on the host cores on a bounded sample: the cylinder
    fixture (1 647 tets, 15 352 dofs), quasi-Newton iterations with one Jacobian, until ~budget_s is spent."""
    from oracle.backend import OracleBackend
    from vasp_amd.monolithic import prepare
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(
            ["-p", "cylinder", "-dt", "0.001", "-T", "0.1", "--theta", "0.501", "--verbose", "False", "--folder",
             tempfile.mkdtemp(), "--sub-folder", "1", "--new-arguments",
             f"mesh_path={ROOT / 'tests' / 'golden' / 'cylinder' / 'cylinder.h5'}"])
    t0 = time.perf_counter()
    ob = OracleBackend(desc)            # includes A_pre = assemble(J_linear), as the reference's solver_setup
    its, step, t = 0, 0, 0.0
    while time.perf_counter() - t0 < budget_s or its == 0:
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
    dt = time.perf_counter() - t0
    ndof = ob.o.ndof
    return {"value": its / dt, "unit": "Newton-iterations/s", "cores": 1, "kind": "port",
            "sample": f"oracle (numpy + SuperLU) on the cylinder fixture: 1647 tets, {ndof} dofs, {step} time steps, "
                      f"{its} Newton iterations in {dt:.1f} s incl. one Jacobian + LU",
            "dof_updates_per_s": its * ndof / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tets", type=int, default=int(os.environ.get("VASPFSI_BENCH_TETS", 1000000)))
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    from vasp_amd.monolithic import prepare

    tmp = Path(tempfile.mkdtemp(prefix=f"vaspfsi_bench_r{rank}_"))
    mesh_path = tmp / "stenosis.h5"
    t_setup = time.perf_counter()
    write_mesh(mesh_path, args.tets, seed=0)
    T_end = args.dt * (args.steps + args.warmup)
    with contextlib.redirect_stdout(io.StringIO()):
        ns, desc, bc_values, pressure, hook = prepare(
            ["-p", "offset_stenosis", "-dt", str(args.dt), "-T", str(T_end), "--theta", "0.501", "--verbose", "False",
             "--folder", str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={mesh_path}"])
    partitioned = world > 1 or force
    if partitioned:
        from vasp_amd.partition import DistBackend
        hb = DistBackend(desc, dist, device=local_rank, lin_max_it=int(os.environ.get("VASPFSI_LIN_MAX_IT", 4000)))
    else:
        hb = HipBackend(desc, device=local_rank)
    ns_cheb = (int(os.environ.get("FSI_CHEB_S", 300)), int(os.environ.get("FSI_CHEB_F", 4)))
    solid_fp32 = int(os.environ.get("FSI_SOLID_FP32", 1)) != 0
    solid_fused = solid_fp32 and int(os.environ.get("FSI_SOLID_BJ", 1)) != 0 and int(os.environ.get("FSI_SOLID_FUSED", 1)) != 0
    mesh = ns["mesh"]
    setup_s = time.perf_counter() - t_setup
    newton = dict(atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=ns["lmbda"], recompute=ns["recompute"],
                  recompute_tstep=ns["recompute_tstep"])

    def one_step(counter, t):
        with contextlib.redirect_stdout(io.StringIO()):
            ns["t"] = t
            hook("pre_solve")(**ns)
        hb.set_dirichlet_values(bc_values())
        hb.set_interface_pressure(float(pressure.P))
        hist = hb.newton_solve(counter=counter, first_step_num=0, **newton)
        hb.shift()
        return hist

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t, counter = 0.0, 0
    for _ in range(args.warmup):
        t += args.dt
        one_step(counter, t)
        counter += 1
    hb.timers(reset=True)
    barrier()
    t0 = time.perf_counter()
    n_newton, n_krylov = 0, 0
    for _ in range(args.steps):
        t += args.dt
        hist = one_step(counter, t)
        counter += 1
        n_newton += len(hist)
        n_krylov += sum(h[3] for h in hist)
    barrier()
    elapsed = time.perf_counter() - t0
    tm = hb.timers()
    wire = "cuda" if (dist is None or dist.get_backend() == "nccl") else "cpu"
    elapsed, total_newton = aggregate(dist, elapsed, n_newton, device=wire)
    if partitioned:
        total_newton /= world        # one partitioned job: every rank counted the same Newton iterations
        _, C_all = aggregate(dist, 0.0, len(hb.part.cells), device=wire)       # cells incl. ghost layers, summed

    if rank == 0:
        ndof, nnz = hb.ndof, int(hb.lib.fsi_matrix_nnz(hb.ctx))
        C = mesh.num_cells
        C_rank = len(hb.part.cells) if partitioned else C       # the kernels of this rank run on its local cells / rows
        ndof_rank = hb.part.ndof if partitioned else ndof
        # algorithmic bytes per launch of each timed kernel
        f_launches = tm["precond_applies"] * ns_cheb[1]                                # fluid velocity sweeps
        sweeps = max(0, tm["inner_vv_iters"] - f_launches)                              # fine-level solid sweeps
        ss_avg = tm["solid_spmv_ms"] / max(1, tm["solid_spmv_calls"])
        db_avg = tm["db_spmv_ms"] / max(1, tm["db_spmv_calls"])
        scalar, tiled = bool(tm["disp_scalar"] & 1), bool(tm["disp_scalar"] & 2)
        db_launches = f_launches + (0 if scalar else tm["inner_dd_iters"])
        sc_avg = tm["sc_spmv_ms"] / max(1, tm["sc_spmv_calls"])
        sc_launches = tm["inner_dd_iters"] if scalar else 0
        # per launch: values (+ 2-byte LDS index or 4-byte column) per pair, the vector once (tiled: once per tile
        # entry, 16 B + 4 B index), row pointers, flags and the float4 result per node
        sc_bytes = (tm["db_pairs"] * 6.0 + tm["tile_entries"] * 20.0 + tm["db_nodes"] * 27.0) if tiled else \
                   (tm["db_pairs"] * 8.0 + tm["db_nodes"] * 43.0)
        dbf_bytes = (tm["db_pairs"] * 14.0 + tm["tile_entries"] * 20.0 + tm["db_nodes"] * 24.0) if tiled else \
                    (tm["db_pairs"] * 16.0 + tm["db_nodes"] * 40.0)
        kernels = {
            ("k_spmv_tiled_f32<1>" if tiled else "k_spmv_sc_f32") + " (displacement block sweeps: one FP32 ratio per node pair"
            + (", neighbour vector entries staged in LDS" if tiled else "") + "; avg from sampled HIP events)":
                (sc_avg * sc_launches, int(sc_launches), sc_bytes),
            ("k_spmv_tiled_f32<3>" if tiled else "k_spmv_db_f32") + " (fluid velocity block sweeps, FP32 component-diagonal node blocks; avg from sampled HIP events)":
                (db_avg * db_launches, int(db_launches), dbf_bytes),
            # (time attributed in the timed region [ms], launches, algorithmic bytes per launch, avg launch [ms])
            (("k_sweep_sb_b3<0> (solid velocity block, fine level: FP32 3x3 block-CSR product fused with the Chebyshev update; avg from sampled HIP events)"
              if solid_fused else "k_spmv_sb (solid velocity block, FP32 3x3 block-CSR, Chebyshev sweeps; avg from sampled HIP events)")
             if solid_fp32 else "k_spmv<1> (solid velocity block, CSR f64, Chebyshev sweeps; avg from sampled HIP events)"):
                (ss_avg * sweeps, int(sweeps),
                 (tm["solid_nnz"] * 4.0 + tm["solid_nnz"] / 9 * 4.0 + tm["solid_rows"] * 8.0 + (tm["solid_rows"] / 3 + 1) * 8.0
                  # fused sweep: per node also r, x (read + write), d_out (write) as float4 and the 3x3 scaling block
                  + (tm["solid_rows"] / 3 * (5 * 16.0 + 48.0) if solid_fused else 0.0)) if solid_fp32
                 else (tm["solid_nnz"] * 12.0 + tm["solid_rows"] * 16.0 + (tm["solid_rows"] + 1) * 8.0)),
            # node-blocked product: the six rows of a node share one column index per entry column (b_i = 4/6 B per entry)
            ("k_spmv_node6 (monolithic Jacobian, f64 values, one i32 column per six entries; + k_spmv<0> on the pressure rows)"
             if not os.environ.get("FSI_SPMV_GENERIC") else "k_spmv<0> (monolithic Jacobian, CSR f64 + i32)"):
                (tm["spmv_ms"], tm["spmv_calls"],
                 nnz * (8.0 + (4.0 / 6.0 if not os.environ.get("FSI_SPMV_GENERIC") else 4.0)) + ndof_rank * 16.0 + (ndof_rank + 1) * 8.0),
            "k_residual (element residual + scatter-add)": (tm["residual_ms"], tm["residual_calls"], C_rank * 1676.0),
            "k_jacobian<nonlinear> (element Jacobian + scatter-add)": (tm["jacobian_ms"], tm["jacobian_calls"], C_rank * 33420.0),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, calls, nbytes = kernels[dom]
        achieved = nbytes / (ms / max(calls, 1) * 1e-3) / 1e9 if ms > 0 else 0.0
        out = {
            "metric": "Newton-iterations/sec (offset_stenosis, monolithic ALE-FSI step)",
            "value": total_newton / elapsed, "unit": "Newton-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"offset_stenosis synthetic mesh, {C} tets, {ndof} dofs, dt={args.dt}, theta=0.501, "
                                   f"quasi-Newton atol=rtol=1e-6 recompute_tstep={ns['recompute_tstep']}",
                       "tets": C, "dofs": ndof, "matrix_nnz": nnz, "parallelism": (f"element partition over {world} ranks (node slabs, ghost-layer cells {C_all / C - 1:.1%}), "
                                       f"halo + all-reduce over {dist.get_backend()}") if partitioned else "1 GPU",
                       "rank0_matrix_nnz": nnz},
            "dof_updates_per_s": total_newton * ndof / elapsed,
            "newton_iterations": n_newton, "krylov_iterations": n_krylov,
            "phase_ms": {k: tm[k] for k in ("residual_ms", "jacobian_ms", "factor_ms", "spmv_ms", "precond_ms", "ortho_ms", "krylov_ms")},
            "phase_calls": {k: tm[k] for k in ("residual_calls", "jacobian_calls", "factor_calls", "spmv_calls", "precond_calls",
                                               "krylov_solves", "krylov_iters", "inner_vv_iters", "inner_schur_iters", "inner_dd_iters")},
            "setup_s": setup_s,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launches": calls,
                         "avg_launch_ms": ms / max(calls, 1), "algorithmic_bytes_per_launch": nbytes},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    hb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
