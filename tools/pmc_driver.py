"""Workload for the rocprofv3 --pmc passes: known-byte calibration streams, then two time steps of the bench problem.
Run as `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/pmc_driver.py` (and again with
WRITE_SIZE); tools/pmc_traffic.py turns the two counter files into per-kernel HBM bytes per launch."""
import contextlib, io, os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from vasp_amd.capi import HipBackend
from vasp_amd.meshgen import write_mesh
from vasp_amd.monolithic import advance, prepare

tets = int(os.environ.get("VASPFSI_BENCH_TETS", 1000000))
tmp = Path(tempfile.mkdtemp(prefix="vaspfsi_pmc_"))
write_mesh(tmp / "stenosis.h5", tets, seed=0)
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, bc_values, pressure, hook = prepare(
        ["-p", "offset_stenosis", "-dt", "0.001", "-T", "0.01", "--theta", "0.501", "--verbose", "False", "--folder",
         str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={tmp / 'stenosis.h5'}"])
hb = HipBackend(desc)
for which, fn in ns["dvp_"].items():
    fn.backend, fn.which = hb, which
CAL_BYTES = 4 << 30
rc = hb.lib.fsi_calibration_streams(hb.ctx, CAL_BYTES)
assert rc == 0
print("calibration bytes", CAL_BYTES, flush=True)
for k in range(int(os.environ.get("PMC_STEPS", 2))):
    with contextlib.redirect_stdout(io.StringIO()):
        hist = advance(ns, hb, bc_values, pressure, hook, 0, out=lambda *a: None)
    ns["counter"] += 1
    print("step", k, [h[3] for h in hist], flush=True)
tm = hb.timers()
print("ortho_q_cols", tm["ortho_q_cols"], "ortho_q_launches", tm["ortho_q_launches"], "ldq", tm["ldq"], "q_elem_bytes", tm["q_elem_bytes"],
      "ortho_z_cols", tm["ortho_z_cols"], "ortho_z_launches", tm["ortho_z_launches"], "ndof", hb.ndof, flush=True)
hb.close()
