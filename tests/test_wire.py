"""N4: the wire side of the reference's telemetry handler, src/mpc_main.cpp:26-36 and :81-222.
Frame parsing and reply framing are host code of the product library (csrc/mpc_wire.cpp): tested here without a GPU
against an independent restatement of the reference's rules; the `gpu` tests replay frames through the device."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O


def _dump_double(x):
    """nlohmann::json 2.1.1 number_float dump (src/utils/json.hpp:8306-8392)"""
    if x == 0:
        return "-0.0" if np.signbit(x) else "0.0"
    s = "%.15g" % x
    return s if any(c in s for c in ".eE") else s + ".0"


def _reply(steer, throttle):
    # std::map key order; NULL is the integer 0 in the build without PLOT_TRAJECTORY (mpc_main.cpp:191-194)
    return '42["steer",{"mpc_x":0,"mpc_y":0,"next_x":0,"next_y":0,"steering_angle":%s,"throttle":%s}]' % (_dump_double(steer), _dump_double(throttle))


def _frame(px, py, psi, mph, steer, throttle, ptsx, ptsy):
    d = {"ptsx": list(map(float, ptsx)), "ptsy": list(map(float, ptsy)), "psi_unity": 4.12, "psi": float(psi), "x": float(px), "y": float(py),
         "steering_angle": float(steer), "throttle": float(throttle), "speed": float(mph)}
    return '42["telemetry",' + json.dumps(d, separators=(",", ":")) + "]"


def test_parse_frames(pkg):
    L = pkg.library()
    t = pkg.MpcWireTelemetry()
    f = _frame(-40.62, 108.73, 3.733651, 41.5, 0.0312, 0.55, [-32.16, -43.49, -61.09, -78.29, -93.05, -107.77], [113.36, 105.94, 92.88, 78.73, 65.34, 50.57]).encode()
    assert L.mpc_wire_parse(f, len(f), C.byref(t)) == 1
    assert (t.x, t.y, t.psi, t.speed, t.steering_angle, t.throttle, t.npts) == (-40.62, 108.73, 3.733651, 41.5, 0.0312, 0.55, 6)
    assert list(t.ptsx)[:6] == [-32.16, -43.49, -61.09, -78.29, -93.05, -107.77] and t.ptsy[5] == 50.57
    # "x" must not be taken from "ptsx", nor "psi" from "psi_unity" (which comes first in this frame)
    for frame, want in ((b'42["telemetry",null]', 0), (b"42", 2), (b'2probe', 2), (b'42["steer",{"a":1}]', 2), (b"", 2),
                        (b'42["telemetry",{"ptsx":[1,2],"ptsy":[1,2],"x":0,"y":0,"psi":0,"speed":1,"steering_angle":0}]', -1),
                        (b'42["telemetry",{"ptsx":[1,2,3],"ptsy":[1,2,3],"x":0,"y":0,"speed":1,"steering_angle":0}]', -1)):
        assert L.mpc_wire_parse(frame, len(frame), C.byref(t)) == want, frame


def test_reply_framing_is_byte_exact(pkg):
    L = pkg.library()
    buf = C.create_string_buffer(256)
    rng = np.random.default_rng(7)
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, 1e-7, -3.25e-5, 0.1, 1 / 3, 0.123456789012345678, 100.0, 1e15, 1e16, 2.5e-310] + list(rng.uniform(-1, 1, 200)) + list(10.0 ** rng.uniform(-12, 3, 100))
    for s in vals:
        for thr in (vals[int(rng.integers(len(vals)))], 1.0):
            n = L.mpc_wire_format_steer(C.c_double(s), C.c_double(thr), buf, 256)
            assert buf.value.decode() == _reply(s, thr) and n == len(buf.value)
            json.loads(buf.value.decode()[2:])                          # and it is valid JSON, as the simulator expects
    assert L.mpc_wire_format_steer(C.c_double(0.5), C.c_double(0.5), buf, 10) < 0
    n = L.mpc_wire_format_manual(buf, 256)
    assert buf.value == b'42["manual",{}]' and n == 15


def _frames_for(pkg, params, waypoints, B, seed):
    """B lake-track situations as simulator frames: RAW telemetry (before the handler's own latency compensation)."""
    tel = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=seed, latency_s=0.0)
    rng = np.random.default_rng(seed)
    rows = np.zeros((6, B))
    rows[0], rows[1], rows[2] = tel["pose"][0], tel["pose"][1], tel["pose"][2] + 2 * np.pi * rng.integers(-1, 2, B)
    rows[3] = np.minimum(tel["pose"][3], 0.8 * params.max_speed) * 3600.0 / 1609.34            # mph, keeps the compensated speed feasible
    rows[4] = -tel["pose"][4]
    rows[5] = rng.uniform(-0.2, 1.0, B)
    frames = [_frame(rows[0, i], rows[1, i], rows[2, i], rows[3, i], rows[4, i], rows[5, i], tel["ptsx"][:, i], tel["ptsy"][:, i]) for i in range(B)]
    return frames, rows, tel


@pytest.mark.gpu
def test_replay_tool_matches_oracle_handler(pkg, golden_dir, waypoints):
    """lib/mpc_replay: two rounds of 48 interleaved connections through the device; replies framed byte-exactly from the
    device's numbers, the numbers themselves against the oracle's handler (incl. each car's previous throttle)."""
    from helpers import TOL_STEER
    cfgname = "config-fast.json"
    params = pkg.params_from_json(os.path.join(golden_dir, cfgname))
    B = 48
    f1, rows1, tel1 = _frames_for(pkg, params, waypoints, B, 71)
    f2, rows2, tel2 = _frames_for(pkg, params, waypoints, B, 72)
    text = "\n".join(f1 + ['42["telemetry",null]'] * B + f2) + "\n"
    exe = os.path.join(os.path.dirname(pkg.library_path()), "mpc_replay")
    p = subprocess.run([exe, os.path.join(golden_dir, cfgname), "--cars", str(B)], input=text, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = p.stdout.splitlines()
    assert len(out) == 3 * B and out[B:2 * B] == ['42["manual",{}]'] * B
    prev = np.zeros(B)
    for rnd, (rows, tel) in enumerate(((rows1, tel1), (rows2, tel2))):
        lines = out[:B] if rnd == 0 else out[2 * B:]
        for i in range(B):
            assert lines[i].startswith('42["steer",')
            d = json.loads(lines[i][2:])[1]
            assert lines[i] == _reply(d["steering_angle"], d["throttle"])                      # framing of exactly these numbers
            assert d["mpc_x"] == 0 and d["next_y"] == 0
            cfg = O.load_config(cfgname)
            st, ref_steer, ref_thr, _ = O.telemetry_handler(cfg, [rows[0, i], rows[1, i], rows[2, i], rows[3, i], rows[4, i], prev[i]],
                                                            list(tel["ptsx"][:, i]), list(tel["ptsy"][:, i]), 0.0)
            if st == 0:
                assert abs(d["steering_angle"] - ref_steer) * params.max_steering < 5 * TOL_STEER
                assert abs(d["throttle"] - ref_thr) < 1e-5
            prev[i] = d["throttle"]


@pytest.mark.gpu
def test_replay_tool_over_tcp(pkg, golden_dir, waypoints):
    """The same replay over one TCP connection (newline-delimited frames on 127.0.0.1): replies come back on the socket and
    are byte-identical to the stdin/stdout mode's."""
    import socket
    import time
    cfgname = "config-fast.json"
    params = pkg.params_from_json(os.path.join(golden_dir, cfgname))
    B = 8
    f1, _, _ = _frames_for(pkg, params, waypoints, B, 73)
    text = "\n".join(f1 + ['42["telemetry",null]'] + f1[:B - 1]) + "\n"
    exe = os.path.join(os.path.dirname(pkg.library_path()), "mpc_replay")
    cfg = os.path.join(golden_dir, cfgname)
    ref = subprocess.run([exe, cfg, "--cars", str(B)], input=text, capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0, ref.stderr
    s0 = socket.socket(); s0.bind(("127.0.0.1", 0)); port = s0.getsockname()[1]; s0.close()
    p = subprocess.Popen([exe, cfg, "--cars", str(B), "--tcp", str(port)], stderr=subprocess.PIPE, text=True)
    try:
        assert "listening" in p.stderr.readline()
        c = socket.create_connection(("127.0.0.1", port), timeout=60)
        c.sendall(text.encode()); c.shutdown(socket.SHUT_WR)
        got = b""
        while True:
            chunk = c.recv(65536)
            if not chunk:
                break
            got += chunk
        c.close()
        assert p.wait(timeout=120) == 0
    finally:
        if p.poll() is None:
            p.kill()
    assert got.decode() == ref.stdout and got.count(b"\n") == 2 * B


@pytest.mark.gpu
def test_host_telemetry_entry_runs_on_the_handles_device_and_reports_errors(pkg, golden_dir, waypoints):
    """mpc_telemetry_batch_host / mpc_wire_telemetry_batch_host: host arrays in and out on the HANDLE's device and stream
    (mpc_handle_device says which), the same numbers as the device entry point; unknown options of the replay tool and
    mixed waypoint counts are refused with a message."""
    import ctypes as C
    import torch
    cfgname = "config-fast.json"
    params = pkg.params_from_json(os.path.join(golden_dir, cfgname))
    B = 40
    _, rows, tel = _frames_for(pkg, params, waypoints, B, 74)
    lib = pkg.library()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    tel6, px, py = c(rows), c(tel["ptsx"]), c(tel["ptsy"])
    with pkg.BatchedMPC(params, B, device=0) as mpc:
        assert lib.mpc_handle_device(mpc._h) == 0
        cmd = np.zeros((2, B)); status = np.zeros(B, dtype=np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)
        assert lib.mpc_telemetry_batch_host(mpc._h, B, B, 6, p(tel6), 0.0, p(px), p(py), p(cmd), p(status)) == 0
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(a.copy()).to(dev)
        d_cmd = torch.zeros((2, B), dtype=torch.float64, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        d_px, d_py, d_tel = t(px), t(py), t(tel6)
        assert lib.mpc_telemetry_batch_device(mpc._h, B, B, 6, C.c_void_p(d_tel.data_ptr()), 0.0, C.c_void_p(d_px.data_ptr()), C.c_void_p(d_py.data_ptr()),
                                              C.c_void_p(d_cmd.data_ptr()), None, C.c_void_p(d_st.data_ptr()), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_cmd.cpu().numpy(), cmd) and np.array_equal(d_st.cpu().numpy(), status)
        assert np.array_equal(px, c(tel["ptsx"]))                                   # the host entry leaves the caller's waypoints alone
        # frames with different waypoint counts in one batch are refused, with a message
        W = pkg.MpcWireTelemetry * 2
        w = W()
        w[0].npts, w[1].npts = 6, 5
        rc = lib.mpc_wire_telemetry_batch_host(mpc._h, 2, w, None, 0.0, p(cmd), p(status))
        assert rc == -1 and b"same number of waypoints" in lib.mpc_last_error()
    exe = os.path.join(os.path.dirname(pkg.library_path()), "mpc_replay")
    r = subprocess.run([exe, os.path.join(golden_dir, cfgname), "--cars"], input="", capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "unknown or incomplete option" in r.stderr
    r = subprocess.run([exe, os.path.join(golden_dir, cfgname), "--bogus", "1"], input="", capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
