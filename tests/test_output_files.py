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


def test_visualization_series_is_streamed_in_segments_and_continues_after_a_restart(tmp_path):
    """A long series never sits in memory as a whole: frames go to <name>.h5, <name>_run_1.h5, ... in bounded segments,
    and the XDMF names file and index per time step - what the reference's output_file_lists reads
    [REF src/vasp/postprocessing/postprocessing_common.py:63-121] (a restarted turtleFSI run leaves the same pattern)."""
    m = FsiMesh.read(GOLDEN / "cylinder" / "cylinder.h5")
    rng = np.random.default_rng(2)
    frame = 8 * 3 * m.num_nodes
    w = VisualizationWriter(tmp_path, m, save_deg=2, segment_bytes=2 * frame + 1)       # two frames per file
    states = [rng.standard_normal(m.num_dofs) for _ in range(5)]
    for k, s in enumerate(states):
        w.write(s, 1e-3 * (k + 1))
        assert sum(len(v) for v in w.frames.values()) <= 3 * 2                           # never more than one segment held
    w.flush()
    files, times, idx = parse_xdmf(tmp_path / "velocity.xdmf")
    assert files == ["velocity.h5", "velocity.h5", "velocity_run_1.h5", "velocity_run_1.h5", "velocity_run_2.h5"]
    assert idx == [0, 1, 0, 1, 0] and np.allclose(times, 1e-3 * np.arange(1, 6))
    for k, (f, i) in enumerate(zip(files, idx)):
        arr = read_h5(tmp_path / f)["VisualisationVector"][str(i)].data
        assert np.array_equal(arr, m.split(states[k])[1])
    assert "Mesh" in read_h5(tmp_path / "velocity.h5") and "Mesh" not in read_h5(tmp_path / "velocity_run_1.h5")
    # --restart-folder: a new writer adopts the series and goes on in the next run file
    w2 = VisualizationWriter(tmp_path, m, save_deg=2, run_index=3)
    extra = rng.standard_normal(m.num_dofs)
    w2.write(extra, 6e-3)
    w2.flush()
    files, times, idx = parse_xdmf(tmp_path / "velocity.xdmf")
    assert files[-1] == "velocity_run_3.h5" and idx[-1] == 0 and len(times) == 6 and times[-1] == 6e-3
    assert np.array_equal(read_h5(tmp_path / "velocity_run_3.h5")["VisualisationVector"]["0"].data, m.split(extra)[1])
