"""Run one problem through the HIP backend and print Newton/Krylov statistics + timers (GPU box helper)."""
import sys, time, io, contextlib, json
sys.path.insert(0, ".")
import numpy as np
from vasp_amd.monolithic import run
from vasp_amd import capi

problem, mesh, dt, T = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
extra = sys.argv[5:]
buf = io.StringIO()
t0 = time.time()
lines = []
def out(s):
    lines.append(s)
with contextlib.redirect_stdout(buf):
    ns = run(["-p", problem, "-dt", dt, "-T", T, "--theta", "0.51", "--folder", "/tmp/run_case", "--sub-folder", "1",
              "--verbose", "False", "--new-arguments", f"mesh_path={mesh}", *extra], out=out)
wall = time.time() - t0
hb = ns["backend"]
print("ndof", hb.ndof, "nnz", hb.lib.fsi_matrix_nnz(hb.ctx), "wall %.1fs" % wall, "loop %.2fs" % ns["time_loop_seconds"],
      "newton its", ns["newton_iterations"])
for step, hist in enumerate(hb.history):
    print("step", step, " | ".join("r=%.2e d=%.2e%s k=%d rr=%.1e" % (h[0], h[1], "*" if h[2] else "", h[3], h[4]) for h in hist))
print(json.dumps(hb.timers()))
txt = buf.getvalue().splitlines()
for l in txt:
    if l.startswith("Probe Point 5") or l.startswith("Probe Point 0"):
        print(l)
for l in lines[-3:]:
    print(l)
