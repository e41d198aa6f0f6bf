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
