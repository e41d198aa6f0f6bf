"""Generates the golden vectors under tests/golden/ with the CPU oracle (numpy).  Run from the repo root:

    python tests/golden/make_golden.py stenosis_ref      # the reference's known-answer run, reference tolerances
    python tests/golden/make_golden.py stenosis_tight    # same run converged to 1e-11 (GPU parity target)
    python tests/golden/make_golden.py cylinder_ref / cylinder_tight

Each run drives ``vasp_amd.monolithic.run`` with ``oracle.backend.OracleBackend`` (exact sparse LU, turtleFSI's
quasi-Newton policy) on the reference's own mesh fixture and stores the state after every time step, the Newton
history and the probe lines.  The reference cannot run here, so these files are outputs of the
restatement; what pins them to the reference are the literal values in tests/test_oracle_pins.py.
"""
import contextlib
import io
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.backend import OracleBackend  # noqa: E402
from vasp_amd.monolithic import run  # noqa: E402

CASES = {
    "stenosis": dict(problem="offset_stenosis", mesh="tests/golden/offset_stenosis/offset_stenosis.h5", dt="0.01", T="0.04"),
    "cylinder": dict(problem="cylinder", mesh="tests/golden/cylinder/cylinder.h5", dt="0.001", T="0.002"),
    # REF tests/test_simulations.py:80-90: the aneurysm problem on its fixture, inlet_id=4 override, 3 steps
    "aneurysm": dict(problem="aneurysm", mesh="tests/golden/aneurysm/small_aneurysm.h5", dt="0.001", T="0.002",
                     more=["inlet_id=4"]),
}


def main(name):
    case, mode = name.rsplit("_", 1)
    c = CASES[case]
    extra = ["--atol", "1e-11", "--rtol", "1e-14"] if mode == "tight" else []
    states = []

    class Recorder(OracleBackend):
        def shift(self):
            super().shift()
            states.append(self.U.copy())

    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        ns = run(["-p", c["problem"], "-dt", c["dt"], "-T", c["T"], "--theta", "0.51", "--folder", f"/tmp/golden_{name}",
                  "--sub-folder", "1"] + extra + ["--new-arguments", f"mesh_path={ROOT / c['mesh']}"] + c.get("more", []),
                 backend_factory=Recorder, out=print)
    log = [l for l in buf.getvalue().splitlines()
           if l.startswith(("Newton", "Probe", "Compute", "Solved", "ramp", "Instant", "  ", "Flow", "Minimum"))]
    hist = ns["backend"].history
    np.savez_compressed(ROOT / "tests" / "golden" / f"{name}.npz", states=np.array(states),
                        residuals=np.array([h[-1][0] for h in hist]), iterations=np.array([len(h) for h in hist]),
                        log=np.array("\n".join(log)))
    print(f"{name}: {time.time() - t0:.0f} s, Newton iterations per step {[len(h) for h in hist]}")


if __name__ == "__main__":
    main(sys.argv[1])
