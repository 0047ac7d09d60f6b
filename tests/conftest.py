import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as G
    p = G.load_package()
    if not os.path.exists(p.library_path()):      # a fresh checkout: compile the HIP extension first (hipcc cross-compiles without a GPU)
        p.build_library()
    return p


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def waypoints(pkg, golden_dir):
    return pkg.scenarios.load_waypoints(os.path.join(golden_dir, "lake_track_waypoints.csv"))


@pytest.fixture(scope="session")
def host_twin():
    """TEST-ONLY CPU build of the device solver core (tests/host_twin)."""
    import ctypes as C
    import subprocess
    d = os.path.join(ROOT, "tests", "host_twin")
    subprocess.check_call(["make", "-s", "-C", d])
    return C.CDLL(os.path.join(d, "libhost_twin.so"))
