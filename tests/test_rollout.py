"""Closed-loop rollout: the src/test.cpp:79-111 pattern (solve, feed step 1 back, repeat)
through mpc_rollout_batch_device, against the oracle run the same way and against the curves digitised from the
reference's own figure examples/10-01-2.png (tests/golden/plot_anchors_10-01-2.json)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from helpers import TEST_CPP, TOL_ACCEL, TOL_STEER, TOL_TRAJ


def _oracle_rollout(cfg, state, coef, ylo, yhi, steps):
    cfg.yaw_low, cfg.yaw_high = float(ylo), float(yhi)
    hist = np.zeros((steps, 9)); worst = 0
    s = list(state)
    for t in range(steps):
        st, o9, _, _, _ = O.mpc_solve(cfg, s, coef)
        hist[t] = o9; worst = max(worst, st)
        s = list(o9[:6])
    return worst, hist


@pytest.mark.gpu
def test_rollout_reproduces_reference_figure(pkg, golden_dir):
    import torch
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"))
    cfg = O.load_config("config-stable.json")
    # run() once (test.cpp:67) -- on the device, then 25 closed-loop solves (test.cpp:82-111)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    B = 64                                   # the same scenario in every lane of one wave
    pose = np.tile(np.array(TEST_CPP["pose"]).reshape(6, 1), (1, B))
    px = np.tile(np.array(TEST_CPP["ptsx"]).reshape(6, 1), (1, B)); py = np.tile(np.array(TEST_CPP["ptsy"]).reshape(6, 1), (1, B))
    with pkg.BatchedMPC(params, B, device=0) as mpc:
        r = mpc.run_torch(t(pose), t(px), t(py), want_pre=True)
        out8 = r["out8"]; pre = r["pre"]
        state = torch.stack([out8[0], out8[1], out8[2], out8[3], out8[6], out8[7]]).contiguous()   # test.cpp:79-80
        coeffs = pre[6:11].contiguous(); ylo = pre[11].contiguous(); yhi = pre[12].contiguous()
        state0 = state.cpu().numpy().copy()
        ro = mpc.rollout_torch(state, coeffs, ylo, yhi, steps=25)
        torch.cuda.synchronize()
        hist = ro["hist"].cpu().numpy(); status = ro["status"].cpu().numpy(); iters = ro["iters"].cpu().numpy()
        final = state.cpu().numpy()
        o8 = out8.cpu().numpy(); pre_h = pre.cpu().numpy()
    assert (status == 0).all() and (iters > 25).all()
    assert np.all(hist == hist[:, :, :1])                       # identical lanes give identical answers
    assert np.array_equal(final, hist[-1, :6])                  # state advanced in place
    # the oracle run the same way
    worst, oh = _oracle_rollout(cfg, state0[:, 0], pre_h[6:11, 0], pre_h[11, 0], pre_h[12, 0], 25)
    assert worst == 0
    assert np.max(np.abs(hist[:, 6, 0] - oh[:, 6])) < 2e-6      # delta, rad (closed loop: errors feed back)
    assert np.max(np.abs(hist[:, :6, 0] - oh[:, :6])) < 2e-5
    # the reference's IPOPT figure
    with open(os.path.join(golden_dir, "plot_anchors_10-01-2.json")) as f:
        anchors = json.load(f)
    pxv, a = anchors["pixel_value"], anchors["curves"]
    cte = np.concatenate([[o8[6, 0]], hist[:, 4, 0]]); epsi = np.concatenate([[o8[7, 0]], hist[:, 5, 0]])
    delta = np.concatenate([[o8[4, 0] * params.max_steering], hist[:, 6, 0]]); v = np.concatenate([[o8[3, 0]], hist[:, 3, 0]])
    assert np.max(np.abs(cte - a["cte"])) <= 2.5 * pxv["cte"]
    assert np.max(np.abs(epsi[5:] - np.array(a["epsi"])[5:])) <= 3.0 * pxv["epsi"]
    assert np.max(np.abs(delta[1:] - np.array(a["delta"])[1:])) <= 2.5 * pxv["delta"]
    assert np.max(np.abs(v - a["v"])) <= 2.0 * pxv["v"]


@pytest.mark.gpu
def test_rollout_batch_matches_oracle(pkg, golden_dir, waypoints):
    import torch
    cfgname = "config-fast.json"
    params = pkg.params_from_json(os.path.join(golden_dir, cfgname))
    cfg = O.load_config(cfgname)
    B, steps = 1024, 6
    sc = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=61)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    with pkg.BatchedMPC(params, B, device=0) as mpc:
        state = t(sc["state"])
        ro = mpc.rollout_torch(state, t(sc["coeffs"]), t(sc["yaw_lo"]), t(sc["yaw_hi"]), steps=steps)
        torch.cuda.synchronize()
        hist = ro["hist"].cpu().numpy(); status = ro["status"].cpu().numpy(); iters = ro["iters"].cpu().numpy()
        # without history: same final state and status
        state2 = t(sc["state"])
        ro2 = mpc.rollout_torch(state2, t(sc["coeffs"]), t(sc["yaw_lo"]), t(sc["yaw_hi"]), steps=steps, want_hist=False)
        torch.cuda.synchronize()
        assert torch.equal(state2, state) and torch.equal(ro2["status"], ro["status"]) and torch.equal(ro2["iters"], ro["iters"])
        st = mpc.stats()
    assert st.batch == B
    assert (status == 0).mean() > 0.9          # psi may leave its yaw bounds in closed loop -> status 3 (infeasible start)
    assert (iters >= steps).all()
    checked = 0
    for i in range(0, B, 32):
        worst, oh = _oracle_rollout(cfg, sc["state"][:, i], sc["coeffs"][:, i], sc["yaw_lo"][i], sc["yaw_hi"][i], steps)
        assert (worst == 0) == (status[i] == 0)
        if worst != 0:
            continue
        checked += 1
        assert np.max(np.abs(hist[:, 6, i] - oh[:, 6])) < TOL_STEER      # every step of the closed loop, default parameters
        assert np.max(np.abs(hist[:, 7, i] - oh[:, 7])) < TOL_ACCEL
        assert np.max(np.abs(hist[:, :6, i] - oh[:, :6])) < TOL_TRAJ
    assert checked >= 24


@pytest.mark.gpu
def test_rollout_argument_checks(pkg, golden_dir):
    import torch
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    dev = torch.device("cuda:0")
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)
    with pkg.BatchedMPC(params, 64, device=0) as mpc:
        with pytest.raises(pkg.MpcError):
            mpc.rollout_torch(z(6, 8), z(5, 8), z(8), z(8), steps=0)
        with pytest.raises(pkg.MpcError):
            mpc.rollout_torch(z(6, 128), z(5, 128), z(128), z(128), steps=1)     # exceeds max_batch


def test_closed_loops_every_solve_against_the_oracle_cpu_build(pkg, host_twin, golden_dir, waypoints):
    """25-step closed loops (src/test.cpp:79-111) of 96 cars, CPU build of the device solver against the oracle, EVERY solve
    compared -- also the ones that start ON a yaw bound (1e-12 inside): with IPOPT's bound_relax_factor (1e-8) restated in both
    solvers that is an ordinary problem (round 2 had to mask 43 % of the solves of such a loop)."""
    from helpers import TOL_ACCEL, TOL_STEER, TOL_TRAJ, closed_loop_report, twin_solve
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    B, steps = 96, 25
    sc = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=122)
    hist = np.zeros((steps, 9, B)); step_status = np.zeros((steps, B), dtype=np.int32)
    st = sc["state"].copy()
    on_bound = 0
    for k in range(steps):
        on_bound += int((np.minimum(sc["yaw_hi"] - st[2], st[2] - sc["yaw_lo"]) < 1e-9).sum())
        r = twin_solve(host_twin, params, dict(state=st, coeffs=sc["coeffs"], yaw_lo=sc["yaw_lo"], yaw_hi=sc["yaw_hi"]), want_traj=False)
        hist[k] = r["out"]; step_status[k] = r["status"]; st = r["out"][:6].copy()
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    _, oh, ost = O.rollout_chunk_full(("config-fast.json", {}, c(sc["state"]), c(sc["coeffs"]), c(sc["yaw_lo"]), c(sc["yaw_hi"]), steps))
    cl = closed_loop_report(hist, step_status, oh, ost)
    cl["cars_that_start_a_solve_on_a_yaw_bound"] = on_bound
    assert on_bound > 200, cl                                     # the case is exercised: hundreds of solves start on their bound
    assert cl["status_differs"] == 0 and cl["cars_on_another_local_minimum"] == 0, cl
    assert cl["d_steer_rad"][3] <= TOL_STEER and cl["d_accel"][3] <= TOL_ACCEL and cl["d_state"][3] <= TOL_TRAJ, cl
