"""The C++ drop-in facade include/mpc_drop_in.hpp (class MPC / Vehicle / RoadGeometry / Config with the
reference's signatures): compiled the way a user of the reference would, run on the reference's own demo
sequence (src/test.cpp:16-111), compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from helpers import TEST_CPP, TOL_ACCEL, TOL_STEER, TOL_TRAJ

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def drop_in_binary(tmp_path_factory, pkg):
    pkg.library()      # the product library must exist
    out = str(tmp_path_factory.mktemp("dropin") / "drop_in_test")
    libdir = os.path.dirname(pkg.library_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(ROOT, "tests", "cpp", "drop_in_test.cpp"), "-L", libdir, "-lmpc_amd",
                           "-Wl,-rpath," + libdir])
    return out


def test_drop_in_compiles_and_fails_loudly_without_gpu(drop_in_binary, golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([drop_in_binary, os.path.join(golden_dir, "config-stable.json"), "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "mpc_create failed" in r.stderr      # no silent CPU path


@pytest.mark.gpu
def test_drop_in_reproduces_test_cpp_sequence(drop_in_binary, golden_dir):
    r = subprocess.run([drop_in_binary, os.path.join(golden_dir, "config-stable.json"), "25"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    run = [float(x) for x in lines[0].split()[1:9]]
    pts = lines[1].split()
    solves = np.array([[float(x) for x in l.split()[2:]] for l in lines[2:]])
    assert solves.shape == (25, 9)
    # oracle: the same sequence
    cfg = O.load_config("config-stable.json")
    st, out8, _, _, pre, _ = O.mpc_run(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    assert st == 0
    ref_run = [out8[0], out8[1], out8[2], out8[3], out8[4] * cfg.max_steering, out8[5], out8[6], out8[7]]
    assert np.max(np.abs(np.array(run) - np.array(ref_run))) < TOL_TRAJ
    assert abs(run[4] - ref_run[4]) < TOL_STEER and abs(run[5] - ref_run[5]) < TOL_ACCEL
    assert float(pts[1]) == pytest.approx(-3.122981, abs=1e-6)          # ptsx transformed in place (MPC.cpp:329)
    assert float(pts[5]) == -0.1 and float(pts[6]) == pytest.approx(0.484345392317, abs=1e-9)   # Config::yawLow/High
    assert "traj 10 2.66806" in lines[0]                                 # trajectory appended, N points
    coef = list(pre.coef)[:pre.nc]
    state = [out8[0], out8[1], out8[2], out8[3], out8[6], out8[7]]
    for i in range(25):
        st, o9, _, _, _ = O.mpc_solve(cfg, state, coef)
        assert st == 0
        assert abs(solves[i, 6] - o9[6]) < TOL_STEER and abs(solves[i, 7] - o9[7]) < TOL_ACCEL
        assert np.max(np.abs(solves[i, :6] - o9[:6])) < TOL_TRAJ
        state = list(o9[:6])
