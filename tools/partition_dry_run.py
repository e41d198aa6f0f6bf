"""CPU-only dry run of what rank 0 does before a partitioned run starts (BASELINE.json configs[4]: 10 M tets
on 8 GPUs): generate the mesh, run the problem file's hooks (``prepare``), cut the global description into every rank's part
(``partition.build_all_parts``) and pickle the parts as the scatter would.  Reports wall time per phase, peak resident memory of
the process, and per rank: owned / ghost cells, dofs, neighbours, halo sizes, message size of its part.

    python tools/partition_dry_run.py [tets=10000000] [ranks=8] > profiles/r05_partition_dry_run_10m_8ranks.txt
"""
import contextlib
import io
import pickle
import resource
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from vasp_amd.meshgen import write_mesh  # noqa: E402
from vasp_amd.monolithic import prepare  # noqa: E402
from vasp_amd.partition import build_all_parts, default_overlap  # noqa: E402

tets = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6          # GB (ru_maxrss is in KiB on Linux)
tmp = Path(tempfile.mkdtemp(prefix="vaspfsi_dry_"))
t0 = time.perf_counter()
m = write_mesh(tmp / "mesh.h5", tets, seed=0)
t1 = time.perf_counter()
print(f"mesh: {len(m['tets'])} tets, {len(m['coords'])} vertices, shape (nc, nr, nx) = {m['shape']}; generated + written in {t1 - t0:.1f} s, "
      f"file {(tmp / 'mesh.h5').stat().st_size / 1e9:.2f} GB, peak RSS {rss():.1f} GB", flush=True)
del m
with contextlib.redirect_stdout(io.StringIO()):
    ns, desc, *_ = prepare(["-p", "offset_stenosis", "-dt", "0.001", "-T", "0.02", "--theta", "0.501", "--verbose", "False", "--folder",
                            str(tmp / "results"), "--sub-folder", "1", "--new-arguments", f"mesh_path={tmp / 'mesh.h5'}"])
t2 = time.perf_counter()
ndof = 6 * int(desc["num_nodes"]) + len(desc["coords"])
print(f"prepare (read, P2 numbering, problem-file hooks): {t2 - t1:.1f} s; {int(desc['num_nodes'])} P2 nodes, {ndof} dofs, peak RSS {rss():.1f} GB", flush=True)
parts = build_all_parts(desc, world)
t3 = time.perf_counter()
print(f"build_all_parts({world} ranks, overlap {default_overlap(world)} node layers): {t3 - t2:.1f} s, peak RSS {rss():.1f} GB", flush=True)
C = len(desc["tet_nodes"])
tot_cells = 0
print("rank   owned cells   ghost cells  ghost/owned   local dofs   owned dofs   ghost dofs(recv)  send dofs   neighbours   part message [MB]")
for q, p in enumerate(parts):
    blob = pickle.dumps(dict(part=p, kw={}), protocol=pickle.HIGHEST_PROTOCOL)
    own, loc = int(p["num_owned_cells"]), len(p["cells"])
    tot_cells += loc
    recv = np.asarray(p["recv_counts"]); send = np.asarray(p["send_counts"])
    print(f"{q:4d} {own:13d} {loc - own:13d} {(loc - own) / max(own, 1):11.1%} {int(p['ndof']) if 'ndof' in p else len(p['l2g_dofs']):12d} "
          f"{int(np.count_nonzero(p['owned_dof_mask'])):12d} {int(recv.sum()):15d} {int(send.sum()):12d} "
          f"{int(np.count_nonzero((recv > 0) | (send > 0))):10d} {len(blob) / 1e6:16.1f}")
t4 = time.perf_counter()
print(f"cells held by all ranks together: {tot_cells} = {tot_cells / C:.3f} x the mesh; pickling all parts: {t4 - t3:.1f} s")
print(f"total wall {t4 - t0:.1f} s, peak RSS of the driver process {rss():.1f} GB (this container: 8 cores, 64 GB)")
