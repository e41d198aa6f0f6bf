"""Result files: layout and XDMF text as VaSP's post-processing tools read them."""
import json
import re

import numpy as np

from conftest import GOLDEN
from vasp_amd.h5lite import read_h5
from vasp_amd.mesh import FsiMesh
from vasp_amd.output import VisualizationWriter, checkpoint, read_checkpoint, refine_topology


def parse_xdmf(path):
    """output_file_lists of the reference [REF src/vasp/postprocessing/postprocessing_common.py:63-121], restated."""
    lines = path.read_text().splitlines()
    ck = any("FiniteElementFunction" in l for l in lines)
    files, times, idx = [], [], []
    for l in lines:
        if "<Time Value" in l:
            times.append(float(re.findall('<Time Value="(.+?)"', l)[0]))
        if ck and "vector" in l:
            files.append(re.findall(r'"HDF">(.*?):', l)[0]); idx.append(int(re.findall(r'_([0-9]+)\/vector', l)[0]))
        elif not ck and "VisualisationVector" in l:
            files.append(re.findall('"HDF">(.+?):/', l)[0]); idx.append(int(re.findall("VisualisationVector/(.+?)</DataItem", l)[0]))
    return files, times, idx


def test_refined_mesh_counts_and_volumes():
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    topo = refine_topology(m)
    assert m.num_nodes == 2500 and len(topo) == 13176                     # REF tests/test_postprocess_mesh.py:33-42
    vol = lambda x: np.abs(np.linalg.det(np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]], axis=2))) / 6
    child = vol(m.node_coords[topo]).reshape(-1, 8)
    assert np.allclose(child.sum(axis=1), vol(m.coords[m.tets]), rtol=1e-12)
    assert np.allclose(child, child[:, :1], rtol=1e-9)                     # red refinement: 8 children of equal volume


def test_visualization_files(tmp_path):
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    rng = np.random.default_rng(0)
    w = VisualizationWriter(tmp_path, m, save_deg=2)
    states = [rng.standard_normal(m.num_dofs) for _ in range(3)]
    for k, s in enumerate(states):
        w.write(s, 1e-3 * (k + 1))
    for name, fld, ncomp in (("displacement", 0, 3), ("velocity", 1, 3), ("pressure", 2, 1)):
        files, times, idx = parse_xdmf(tmp_path / f"{name}.xdmf")
        assert files == [f"{name}.h5"] * 3 and idx == [0, 1, 2] and np.allclose(times, [1e-3, 2e-3, 3e-3])
        g = read_h5(tmp_path / f"{name}.h5")
        assert g["Mesh"]["0"]["mesh"]["geometry"].data.shape == (2500, 3)
        assert g["Mesh"]["0"]["mesh"]["topology"].data.shape == (13176, 4)
        for k in range(3):
            arr = g["VisualisationVector"][str(k)].data
            assert arr.shape == (2500, ncomp)
            ref = m.split(states[k])[fld]
            if ncomp == 3:
                assert np.array_equal(arr, ref)                            # node 0 of the file = mesh vertex 0
            else:
                assert np.array_equal(arr[:m.num_vertices, 0], ref)
    w1 = VisualizationWriter(tmp_path / "deg1", m, save_deg=1)
    w1.write(states[0], 0.0)
    g = read_h5(tmp_path / "deg1" / "velocity.h5")
    assert g["VisualisationVector"]["0"].data.shape == (352, 3) and g["Mesh"]["0"]["mesh"]["topology"].data.shape == (1647, 4)


def test_checkpoint_round_trip(tmp_path):
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    x = np.random.default_rng(1).standard_normal(m.num_dofs)
    checkpoint(tmp_path, m, x, dict(dt=1e-3, theta=0.51, folder="f", save_deg=2, dx_f_id=1, dx_s_id=2, mu_f=3.5e-3), 0.02, 19)
    meta = json.loads((tmp_path / "default_variables.json").read_text())
    assert meta["t"] == 0.02 and meta["counter"] == 19 and meta["save_deg"] == 2          # keys the post tools read
    assert np.array_equal(read_checkpoint(tmp_path, m), x)
    files, times, idx = parse_xdmf(tmp_path / "checkpoint_v1.xdmf")
    assert files == ["checkpoint_v1.h5"] and idx == [0] and times == [0.02]
    g = read_h5(tmp_path / "checkpoint_v1.h5")["v1"]["v1_0"]
    assert sorted(g) == ["cell_dofs", "cells", "mesh", "vector", "x_cell_dofs"]
    assert g["vector"].data.shape == (3 * m.num_nodes, 1) and g["x_cell_dofs"].data.shape == (m.num_cells + 1,)
    assert g["cell_dofs"].data.shape == (30 * m.num_cells,) and not (tmp_path / "tmp_checkpoint_v1.h5").exists()


def last_frame_as_predeform_mesh_reads_it(h5_path):
    """REF src/vasp/postprocessing/predeform_mesh.py:55-57, restated without h5py: the displacement of "the last time
    step" is dataset number len(keys) - 1 of /VisualisationVector in displacement.h5."""
    g = read_h5(h5_path)["VisualisationVector"]
    number_of_datasets = len(g.keys())
    return g[f"{number_of_datasets - 1}"].data[:, :]


def frames_as_create_hdf5_reads_them(viz_folder, field):
    """REF src/vasp/postprocessing/postprocessing_fenics/create_hdf5.py:89-98,139-160, restated: the XDMF gives (file,
    index) per time step; the first file is opened for VisualisationVector/0, then every step opens the file its entry
    names and reads VisualisationVector/<index>."""
    files, times, idx = parse_xdmf(viz_folder / f"{field}.xdmf")
    first = read_h5(viz_folder / files[0])["VisualisationVector"]["0"].data[:, :]
    out = [read_h5(viz_folder / f)["VisualisationVector"][str(i)].data[:, :] for f, i in zip(files, idx)]
    return first, times, out


def test_every_frame_of_a_run_is_appended_to_one_file_and_the_series_continues_after_a_restart(tmp_path):
    """DOLFIN appends every frame to <name>.h5 and the consumers rely on it.  40 frames go
    through the writer one at a time (nothing but the current frame is held); after EVERY append the file on disk is a
    complete HDF5 file whose last dataset is the frame just written - the read of vasp-predeform-mesh - and the XDMF walk
    of vasp-create-hdf5 returns every frame; a walker that shares no code with h5lite sees the same tree."""
    from test_h5_structure_independent import Walker
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    rng = np.random.default_rng(2)
    w = VisualizationWriter(tmp_path, m, save_deg=2)
    states = []
    for k in range(40):
        states.append(rng.standard_normal(m.num_dofs))
        w.write(states[-1], 1e-3 * (k + 1))
        assert not hasattr(w, "frames")                                                  # no frame store in the writer
        if k in (0, 1, 7, 8, 9, 31, 32, 39):                                             # across symbol-node / B-tree growth
            assert np.array_equal(last_frame_as_predeform_mesh_reads_it(tmp_path / "displacement.h5"), m.split(states[k])[0])
    first, times, frames = frames_as_create_hdf5_reads_them(tmp_path, "velocity")
    assert np.allclose(times, 1e-3 * np.arange(1, 41)) and np.array_equal(first, m.split(states[0])[1])
    for k, fr in enumerate(frames):
        assert np.array_equal(fr, m.split(states[k])[1])
    files, _, idx = parse_xdmf(tmp_path / "pressure.xdmf")
    assert files == ["pressure.h5"] * 40 and idx == list(range(40))                      # ONE file per field
    assert sorted(p.name for p in tmp_path.glob("*.h5")) == ["displacement.h5", "pressure.h5", "velocity.h5"]
    # independent structural walk: group entries name-sorted (the format's lookup order), raw data intact, EOF = file length
    wk = Walker(tmp_path / "velocity.h5")
    tree = wk.tree()
    vec = [p for p in tree if p.startswith("/VisualisationVector/")]
    assert vec == sorted(vec) and len(vec) == 40 and wk.super["eof_is_len"]
    assert tree["/Mesh/0/mesh/topology"]["attributes"] and tree["/VisualisationVector"]["message_types"] == [0x0011]
    for k in (0, 17, 39):
        assert np.frombuffer(tree[f"/VisualisationVector/{k}"]["raw"], dtype="<f8").reshape(-1, 3).tolist() == m.split(states[k])[1].tolist()
    w.close()
    # --restart-folder: a new writer adopts the series and goes on in <name>_run_<N>.h5 behind the same XDMF (the pattern
    # a restarted turtleFSI run leaves, resolved per time step by output_file_lists)
    w2 = VisualizationWriter(tmp_path, m, save_deg=2, run_index=3)
    extra = [rng.standard_normal(m.num_dofs) for _ in range(2)]
    for k, s_ in enumerate(extra):
        w2.write(s_, 1e-3 * (41 + k))
    w2.flush()
    files, times, idx = parse_xdmf(tmp_path / "velocity.xdmf")
    assert files[-2:] == ["velocity_run_3.h5"] * 2 and idx[-2:] == [0, 1] and len(times) == 42 and times[-1] == 42e-3
    g3 = read_h5(tmp_path / "velocity_run_3.h5")
    assert np.array_equal(g3["VisualisationVector"]["1"].data, m.split(extra[1])[1]) and "Mesh" in g3
    _, _, frames = frames_as_create_hdf5_reads_them(tmp_path, "velocity")
    assert len(frames) == 42 and np.array_equal(frames[-1], m.split(extra[1])[1]) and np.array_equal(frames[3], m.split(states[3])[1])


def test_checkpoint_files_under_the_independent_walker(tmp_path):
    """Checkpoint/checkpoint_{d1,v1,p1}.h5 in DOLFIN's write_checkpoint layout
    [REF src/vasp/postprocessing/postprocessing_h5py/postprocessing_h5py_common.py:235-241,639-670], walked by the parser
    that shares no code with h5lite: names, shapes, element types."""
    from test_h5_structure_independent import Walker
    import struct
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    x = np.random.default_rng(3).standard_normal(m.num_dofs)
    checkpoint(tmp_path, m, x, dict(dt=1e-3), 0.01, 9)
    C, N2, V = m.num_cells, m.num_nodes, m.num_vertices
    for name, nvec, ndofs in (("d1", 3 * N2, 30 * C), ("v1", 3 * N2, 30 * C), ("p1", V, 4 * C)):
        wk = Walker(tmp_path / f"checkpoint_{name}.h5")
        tree = wk.tree()
        base = f"/{name}/{name}_0"
        assert sorted(p for p in tree if tree[p]["kind"] == "dataset") == sorted(
            f"{base}/{d}" for d in ("cell_dofs", "cells", "mesh/geometry", "mesh/topology", "vector", "x_cell_dofs"))
        dims = lambda path: struct.unpack_from(f"<{tree[path]['dataspace'][1]}Q", tree[path]["dataspace"], 8)
        assert dims(f"{base}/vector") == (nvec, 1) and dims(f"{base}/cell_dofs") == (ndofs,)
        assert dims(f"{base}/x_cell_dofs") == (C + 1,) and dims(f"{base}/cells") == (C,)
        assert dims(f"{base}/mesh/geometry") == (V, 3) and dims(f"{base}/mesh/topology") == (C, 4)
        assert tree[f"{base}/vector"]["datatype"][0] & 0x0F == 1 and tree[f"{base}/cell_dofs"]["datatype"][0] & 0x0F == 0   # f64 / i64
        assert wk.super["eof_is_len"]
