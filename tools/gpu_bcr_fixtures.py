import sys, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import GOLDEN, prepare_case
from vasp_amd.capi import HipBackend
import numpy as np, contextlib, io
cases = [("cylinder", GOLDEN/"cylinder"/"cylinder.h5", {}), ("offset_stenosis", GOLDEN/"offset_stenosis"/"offset_stenosis.h5", dict(dt="0.01", T="0.04")),
         ("aneurysm", GOLDEN/"aneurysm"/"small_aneurysm.h5", dict(extra=("inlet_id=4",))), ("predeform", GOLDEN/"cylinder"/"cylinder.h5", dict(dt="0.01", T="0.02", theta="1.0"))]
for name, mesh, kw in cases:
    case = prepare_case(name, mesh, tempfile.mkdtemp(), **kw)
    ns, desc, bc_values, pressure, hook = case
    hb = HipBackend(desc)
    with contextlib.redirect_stdout(io.StringIO()):
        ns["t"] = float(ns["dt"]); hook("pre_solve")(**ns)
    hb.set_dirichlet_values(bc_values()); hb.set_interface_pressure(float(pressure.P) if pressure is not None else 0.0)
    hb.assemble_residual(); hb.assemble_jacobian()
    i = hb.solid_coarse_info()
    print(f"{name:16s} {mesh.name:22s} coarse nodes {i['nodes']:6d}  planned {i['planned']} ready {i['ready']}  BFS blocks {i['bfs_blocks']:4d}  largest {i['max_block']:5d}  levels {i['levels']}  operators {i['operator_bytes']/1e6:7.1f} MB  launches {i['launches_per_solve']}")
    hb.close()
