"""Round 4, VERDICT r3 item 1b: how far the shipped Newton policy lands from the exact-solve trajectory on the known-answer
case (offset-stenosis fixture, 5 steps, tolerances 1e-6) as a function of the forcing terms, with the Krylov cost.

    python tools/gpu_r4_forcing_scan.py > gpurun_out/r04_forcing_scan.txt
"""
import contextlib
import io
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import tempfile
    from conftest import GOLDEN, prepare_case
    from vasp_amd.capi import HipBackend
    case = prepare_case("offset_stenosis", GOLDEN / "offset_stenosis" / "offset_stenosis.h5", tempfile.mkdtemp(), dt="0.01", T="0.04")
    ns, desc, bc_values, pressure, hook = case
    mesh = ns["mesh"]
    N2 = mesh.num_nodes

    def run(env=None, **kw):
        old = {}
        for k, v in (env or {}).items():
            old[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            hb = HipBackend(desc, **kw)
            its, kry = [], 0
            for k in range(5):
                with contextlib.redirect_stdout(io.StringIO()):
                    ns["t"] = 0.01 * (k + 1)
                    hook("pre_solve")(**ns)
                hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P))
                h = hb.newton_solve(counter=k, first_step_num=0, atol=ns["atol"], rtol=ns["rtol"], max_it=ns["max_it"], lmbda=1.0,
                                    recompute=ns["recompute"], recompute_tstep=ns["recompute_tstep"])
                its.append(len(h)); kry += sum(it[3] for it in h)
                hb.shift()
            U = hb.get_state("n")
            tm = hb.timers()
            hb.close()
            return U, its, kry, tm
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    exact_env = {"FSI_KRYLOV_FP32": "0", "FSI_OPERATOR_FP32": "0", "FSI_SCHUR_FP32": "0", "FSI_SWEEPS_FP16": "0"}
    U_ref, its_ref, kry_ref, _ = run(exact_env, lin_rtol=1e-10, newton_forcing=0.0)
    print("exact solves: Newton", its_ref, "Krylov", kry_ref)
    configs = [("forcing 1e-2, one stream (round 3)", {"FSI_PREC_STREAMS": "0"}),
               ("forcing 1e-2, two streams", {}),
               ("late 3e-3, factor 10", {"FSI_NEWTON_FORCING_LATE": "3e-3", "FSI_NEWTON_LATE_FACTOR": "10"}),
               ("late 3e-3, factor 30", {"FSI_NEWTON_FORCING_LATE": "3e-3", "FSI_NEWTON_LATE_FACTOR": "30"}),
               ("late 2e-3, factor 10", {"FSI_NEWTON_FORCING_LATE": "2e-3", "FSI_NEWTON_LATE_FACTOR": "10"}),
               ("late 1e-3, factor 10", {"FSI_NEWTON_FORCING_LATE": "1e-3", "FSI_NEWTON_LATE_FACTOR": "10"}),
               ("late 1e-3, factor 100", {"FSI_NEWTON_FORCING_LATE": "1e-3", "FSI_NEWTON_LATE_FACTOR": "100"}),
               ("late 1e-4, factor 10", {"FSI_NEWTON_FORCING_LATE": "1e-4", "FSI_NEWTON_LATE_FACTOR": "10"}),
               ("late 1e-4, factor 100", {"FSI_NEWTON_FORCING_LATE": "1e-4", "FSI_NEWTON_LATE_FACTOR": "100"}),
               ("forcing 3e-3", {"FSI_NEWTON_FORCING": "3e-3"}),
               ("forcing 1e-3", {"FSI_NEWTON_FORCING": "1e-3"})]
    for name, env in configs:
        U, its, kry, tm = run(env)
        errs = {f: float(np.linalg.norm(U[sl] - U_ref[sl]) / np.linalg.norm(U_ref[sl]))
                for f, sl in (("d", slice(0, 3 * N2)), ("v", slice(3 * N2, 6 * N2)), ("p", slice(6 * N2, None)))}
        print(f"{name:32s} Newton {its} Krylov {kry:4d} late solves {tm['newton_late_solves']:2d} Q FP{8 * tm['q_elem_bytes']}  "
              f"d {errs['d']:.2e} v {errs['v']:.2e} p {errs['p']:.2e}", flush=True)


if __name__ == "__main__":
    main()
