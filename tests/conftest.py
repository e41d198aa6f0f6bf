"""Test configuration: the ``gpu`` marker (tests that need an MI355X) and shared fixtures."""
import contextlib
import io
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def prepare_case(problem, mesh, tmp, dt="0.001", T="0.002", theta="0.51", extra=()):
    """Run the host driver up to (not including) the time loop; returns (ns, desc, bc_values, pressure, hook)."""
    from vasp_amd.monolithic import prepare
    with contextlib.redirect_stdout(io.StringIO()):
        return prepare(["-p", problem, "-dt", dt, "-T", T, "--theta", theta, "--verbose", "False", "--folder", str(tmp),
                        "--sub-folder", "1", "--new-arguments", f"mesh_path={mesh}", *extra])


@pytest.fixture(scope="session")
def cylinder_case(tmp_path_factory):
    return prepare_case("cylinder", GOLDEN / "cylinder" / "cylinder.h5", tmp_path_factory.mktemp("cyl"))


@pytest.fixture(scope="session")
def stenosis_case(tmp_path_factory):
    return prepare_case("offset_stenosis", GOLDEN / "offset_stenosis" / "offset_stenosis.h5",
                        tmp_path_factory.mktemp("os"), dt="0.01", T="0.04")


def make_avf_case(tmp_path):
    """The avf problem [REF src/vasp/simulations/avf.py] on a synthetic tube: the reference tree holds neither the AVF mesh
    nor avf.csv, so the downstream half of a generated offset-stenosis tube carries the vein ids (1002 / 1011 / 1022 /
    1033) and the patient table is a short ramp.  Returns prepare()'s tuple."""
    import json
    import numpy as np
    from vasp_amd.mesh import FsiMesh
    from vasp_amd.meshgen import generate
    m = generate(6000)
    x_c = m["coords"][m["tets"]].mean(axis=1)[:, 0]
    x_f = m["coords"][m["facets"]].mean(axis=1)[:, 0]
    cm, fm = m["cell_markers"].copy(), m["facet_markers"].copy()
    mid = 0.008
    cm[(cm == 2) & (x_c > mid)] = 1002
    for a, b in ((11, 1011), (22, 1022), (33, 1033)):
        fm[(fm == a) & (x_f > mid)] = b
    mesh = FsiMesh.from_arrays(m["coords"], m["tets"], cm, m["facets"], fm)
    tmp_path.mkdir(parents=True, exist_ok=True)
    mesh.write(tmp_path / "avf.h5")
    (tmp_path / "avf_probe_point.json").write_text(json.dumps([[0.0, 0.0, 0.0], [16.0, 0.0, 0.0]]))       # mm
    (tmp_path / "avf.csv").write_text("v_PA,v_DA,PV\n" + "\n".join(f"{0.3 + 0.01 * i},{0.1 + 0.005 * i},{9000 + 50 * i}" for i in range(20)))
    return prepare_case("avf", tmp_path / "avf.h5", tmp_path / "run", dt="0.0001", T="0.2", theta="0.501",
                        extra=(f"patient_data_path={tmp_path / 'avf.csv'}", "fsi_region=[0.008,0,0,0.006]"))
