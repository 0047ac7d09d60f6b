"""The solver header (both precisions) and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU.
GPU sanitizers are not available on the pool (no xnack+ builds), so this is where out-of-bounds indexing of the
per-stage records, uninitialised reads and signed overflow in the solver logic would show."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sanitized_exe():
    d = tempfile.mkdtemp()
    exe = os.path.join(d, "twin_sanitize")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "carnd-mpc-project_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle")]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.check_call(["gcc", "-std=gnu99", "-c"] + san + inc + [os.path.join(ROOT, "oracle", "mpc_oracle.c"), "-o", os.path.join(d, "oracle.o")])
    subprocess.check_call(["g++", "-std=c++17"] + san + inc + [os.path.join(ROOT, "tests", "cpp", "twin_sanitize.cpp"),
                                                               os.path.join(ROOT, "carnd-mpc-project_amd", "csrc", "mpc_params.cpp"),
                                                               os.path.join(d, "oracle.o"), "-lm", "-o", exe])
    return exe


@pytest.mark.parametrize("N,dt,B,weights", [(10, 0.1, 48, False), (10, 0.1, 32, True), (25, 0.05, 12, False), (3, 0.1, 16, False), (64, 0.02, 2, False)])
def test_solver_and_oracle_under_asan_ubsan(pkg, golden_dir, waypoints, sanitized_exe, N, dt, B, weights):
    cfg = os.path.join(golden_dir, "config-fast.json")
    params = pkg.params_from_json(cfg, N=N, dt=dt)
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=90 + N)
    rows = [b["state"], b["coeffs"], b["yaw_lo"][None], b["yaw_hi"][None]]
    if weights:
        rows.append(pkg.scenarios.weight_sweep(B, params, seed=91))
    blob = struct.pack("<iiid", N, B, int(weights), dt) + np.ascontiguousarray(np.concatenate(rows, axis=0), dtype=np.float64).tobytes()
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(blob)
    try:
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        p = subprocess.run([sanitized_exe, cfg, f.name], capture_output=True, text=True, env=env, timeout=600)
    finally:
        os.unlink(f.name)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "status_mismatch 0" in p.stdout and "ERROR" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
