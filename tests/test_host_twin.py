"""The device solver core, compiled for the CPU by tests/host_twin (TEST-ONLY), against the oracle.

This is how the interior-point / Riccati logic of carnd-mpc-project_amd/csrc/mpc_core.h is verified in
the GPU-less build container; the same comparisons run on the real HIP path in test_gpu_parity.py."""
import os

import numpy as np
import pytest

import oracle_lib as O
from helpers import TEST_CPP, assert_parity, load_golden, oracle_solve_batch, twin_solve


def test_twin_test_cpp_scenario(pkg, host_twin, golden_dir):
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"))
    cfg = O.load_config("config-stable.json")
    pre, _, _ = O.run_pre(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    coef = np.zeros((5, 1)); coef[:pre.nc, 0] = list(pre.coef)[:pre.nc]
    b = {"state": np.array(list(pre.state)).reshape(6, 1), "coeffs": coef,
         "yaw_lo": np.array([pre.yaw_low]), "yaw_hi": np.array([pre.yaw_high])}
    r = twin_solve(host_twin, params, b)
    assert r["status"][0] == 0
    assert r["out"][6, 0] == pytest.approx(0.0024133755, abs=5e-9)
    assert r["out"][8, 0] == pytest.approx(6243.44368073, abs=1e-5)
    ref = oracle_solve_batch(cfg, b, [0])
    assert_parity(r["out"], ref["out"], r["traj"], ref["traj"], "test.cpp")
    assert abs(int(r["iters"][0]) - int(ref["iters"][0])) <= 2


@pytest.mark.parametrize("config,kind", [("config-fast.json", "lake"), ("config-stable.json", "straight"),
                                         ("config-no-latency.json", "lake")])
def test_twin_matches_oracle(pkg, host_twin, golden_dir, waypoints, config, kind):
    params = pkg.params_from_json(os.path.join(golden_dir, config))
    B = 96
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=21) if kind == "lake" else pkg.scenarios.straight_line_batch(B, params, seed=22)
    r = twin_solve(host_twin, params, b)
    assert (r["status"] == 0).all()
    ref = oracle_solve_batch(O.load_config(config), b, range(B))
    assert (ref["status"] == 0).all()
    assert_parity(r["out"], ref["out"], r["traj"], ref["traj"], "%s/%s" % (config, kind))


def test_twin_long_horizon(pkg, host_twin, golden_dir, waypoints):
    """BASELINE.json configs[3]: N=25, dt=0.05."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"), N=25, dt=0.05)
    b = pkg.scenarios.lake_track_batch(24, params, waypoints, seed=23)
    r = twin_solve(host_twin, params, b)
    assert (r["status"] == 0).all()
    cfg = O.load_config("config-stable.json", N=25, dt=0.05)
    ref = oracle_solve_batch(cfg, b, range(24))
    # default parameters (tol = 1e-8 with the termination polish): the stated 1e-6 on a0 holds, interior a0 included
    assert_parity(r["out"], ref["out"], r["traj"], ref["traj"], "N=25")


def test_termination_polish_pins_weakly_determined_outputs(pkg, host_twin, golden_dir, waypoints):
    """IPOPT's stopping rule (polish = 0) leaves an interior a0 up to ~1e-4 from the limit point: shown here by
    solving the same instances at two tolerances.  With the polish the answers agree to the stated 1e-6 whatever
    iterate crossed the tolerance first -- and so do the far ends of the predicted trajectories (1e-5 m), which move by up
    to 1e-4 m under IPOPT's rule -- at a cost of less than one iteration per solve."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"), N=25, dt=0.05)
    b = pkg.scenarios.lake_track_batch(2048, params, waypoints, seed=42)
    exact = params.copy(); exact.out_step_tol = 1e-14                  # polish to the limit point (6 extra steps at most)
    re_ = twin_solve(host_twin, exact, b)
    plain = params.copy(); plain.polish = 0
    rp = twin_solve(host_twin, plain, b)
    r = twin_solve(host_twin, params, b)
    ok = (re_["status"] == 0) & (rp["status"] == 0) & (r["status"] == 0)
    assert ok.sum() >= 2040            # polishing to 1e-14 sits on the rounding floor: a few line searches give up there
    assert np.max(np.abs(rp["out"][7] - re_["out"][7])[ok]) > 1e-5      # the slack the polish removes
    assert np.max(np.abs(r["out"][7] - re_["out"][7])[ok]) < 1e-7
    assert np.max(np.abs(r["out"][6] - re_["out"][6])[ok]) < 1e-7
    assert np.max(np.abs(rp["traj"] - re_["traj"])[:, ok]) > 5e-5 and np.max(np.abs(r["traj"] - re_["traj"])[:, ok]) < 5e-6
    assert r["iters"][ok].mean() - rp["iters"][ok].mean() < 0.9          # N = 25; 0.43 on the N = 10 headline workload


def test_initial_state_rows_follow_the_oracle_iteration_for_iteration(pkg, host_twin, golden_dir, waypoints):
    """MpcParams.initial_state_rows (include/mpc_amd.h).  The reference's NLP keeps the initial state as six variables pinned by six
    equality rows (MPC.cpp:116-121, 269-281); their multipliers and the bound duals of psi_0 / v_0 decouple from the Newton step,
    so the device solver leaves them out by default -- and its dual infeasibility then lacks the residual those variables' rows keep
    after a step the fraction-to-the-boundary rule has cut: on a few per cent of a batch its barrier parameter comes down an
    iteration before the oracle's (same solution at the end).  With the rows carried the solver takes the oracle's iteration count
    on all but 1-2 % of the instances (those differ by the rounding of the last step); every status and every point as before."""
    import oracle_lib as O
    from helpers import oracle_solve_batch
    for sweep in (False, True):
        params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
        B = 768
        b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=77)
        w = pkg.scenarios.weight_sweep(B, params, seed=78, velocity_weights=(0.0, 1.0, 100.0)) if sweep else None
        ref = oracle_solve_batch(O.load_config("config-fast.json"), b, range(B), weights=w)
        same = {}
        for rows in (0, 1):
            q = params.copy(); q.initial_state_rows = rows
            r = twin_solve(host_twin, q, b, weights=w)
            assert np.array_equal(r["status"], ref["status"])
            ok = r["status"] == 0
            far = np.abs(r["out"][6] - ref["out"][6]) > 1e-6                  # (forks of a flat objective: velocity weight 0)
            assert (ok & far).sum() <= (B // 200 if sweep else 0)
            assert np.max(np.abs(r["out"][:8] - ref["out"][:8])[:, ok & ~far]) < 1e-6
            same[rows] = float((r["iters"] == ref["iters"])[ok].mean())
        assert same[1] >= 0.975 and same[1] >= same[0] + 0.01, same


def test_instances_that_leave_the_central_path(pkg, host_twin, golden_dir, waypoints):
    """Where IPOPT would enter its restoration phase (src/control/MPC.cpp:290-292) and where the iteration cap strikes:
    named instances of the unfiltered draw.  The device solver (CPU build) and the oracle report the SAME status, the same
    iteration count and the same point -- the point is the last iterate, which is what the reference returns as well
    (it prints the status and returns solution.x, MPC.cpp:295-303)."""
    from helpers import OFF_PATH_BATCH, OFF_PATH_INSTANCES
    params = pkg.params_from_json(os.path.join(golden_dir, OFF_PATH_BATCH["config"]))
    b = pkg.scenarios.lake_track_batch(OFF_PATH_BATCH["B"], params, waypoints, seed=OFF_PATH_BATCH["seed"], filtered=False)
    r = twin_solve(host_twin, params, b, want_traj=False)
    counts = np.bincount(r["status"], minlength=5)
    assert list(counts) == [16380, 1, 3, 0, 0]                       # every instance accounted for
    idx = sorted(OFF_PATH_INSTANCES)
    ref = oracle_solve_batch(O.load_config(OFF_PATH_BATCH["config"]), b, idx, opt=O.default_options(max_iter=params.max_iter))
    for j, i in enumerate(idx):
        st, it = OFF_PATH_INSTANCES[i]
        assert (r["status"][i], r["iters"][i]) == (st, it), (i, r["status"][i], r["iters"][i])
        assert ref["status"][j] == st and ref["iters"][j] == it, (i, ref["status"][j], ref["iters"][j])
        assert np.max(np.abs(r["out"][:8, i] - ref["out"][:8, j])) < 1e-6, i


def test_long_horizon_instances_that_leave_the_central_path(pkg, host_twin, golden_dir):
    """helpers.OFF_PATH_N25: the 4 of 262 144 N = 25 instances that end in LINESEARCH, identically in oracle and device solver,
    and the one on the rounding floor of tol, which both report as converged at the same point."""
    from helpers import OFF_PATH_N25 as T
    params = pkg.params_from_json(os.path.join(golden_dir, T["config"]), N=T["N"], dt=T["dt"])
    rows = T["rows"]
    b = {"state": rows[:, :6].T.copy(), "coeffs": rows[:, 6:11].T.copy(), "yaw_lo": rows[:, 11].copy(), "yaw_hi": rows[:, 12].copy()}
    r = twin_solve(host_twin, params, b, want_traj=False)
    ref = oracle_solve_batch(O.load_config(T["config"], N=T["N"], dt=T["dt"]), b, range(len(rows)), opt=O.default_options(max_iter=params.max_iter))
    for i, (st, it) in enumerate(T["expect"]):
        assert r["status"][i] == st and ref["status"][i] == st, (i, r["status"][i], ref["status"][i])
        if it is not None:
            assert r["iters"][i] == it and ref["iters"][i] == it
        assert np.max(np.abs(r["out"][:8, i] - ref["out"][:8, i])) < 1e-6


def test_twin_weight_sweep_with_zero_velocity_weight(pkg, host_twin, golden_dir, waypoints):
    """SURVEY.md section 8d, Config 5: the sweep including velocity weight 0 (the acceleration is then bang-bang between
    maxDeceleration and maxAcceleration, examples/velocity-weights.png): fp64 agrees with the oracle to the stated 1e-6."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    B = 96
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=26)
    w = pkg.scenarios.weight_sweep(B, params, seed=27, velocity_weights=(0.0, 1.0, 100.0))
    assert (w[2] == 0).sum() > 16
    r = twin_solve(host_twin, params, b, weights=w)
    ref = oracle_solve_batch(O.load_config("config-fast.json"), b, range(B), weights=w)
    assert (r["status"] == 0).all() and (ref["status"] == 0).all()
    assert_parity(r["out"], ref["out"], r["traj"], ref["traj"], "weights incl. w_v = 0")
    a0 = r["out"][7, w[2] == 0]
    assert (a0 < params.max_deceleration + 1e-3).any()          # "the vehicle decelerates"


def test_twin_per_instance_weights(pkg, host_twin, golden_dir, waypoints):
    """BASELINE.json configs[4]: per-instance Config::weights."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    B = 48
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=24)
    w = pkg.scenarios.weight_sweep(B, params, seed=25)
    r = twin_solve(host_twin, params, b, weights=w)
    assert (r["status"] == 0).all()
    ref = oracle_solve_batch(O.load_config("config-fast.json"), b, range(B), weights=w)
    assert_parity(r["out"], ref["out"], r["traj"], ref["traj"], "weights")
    # acceleration weight w[6] has provably no effect under the frozen tape
    w2 = w.copy(); w2[6] = 12345.0
    r2 = twin_solve(host_twin, params, b, weights=w2)
    assert np.array_equal(r2["out"], r["out"])


def test_twin_against_scipy_goldens(pkg, host_twin, golden_dir):
    gold = load_golden("scipy_cross_solve.json")
    for cfgname in ("config-stable.json", "config-fast.json"):
        cases = [c for c in gold["cases"] if c["config"] == cfgname]
        params = pkg.params_from_json(os.path.join(golden_dir, cfgname))
        b = {"state": np.array([c["state"] for c in cases]).T.copy(), "coeffs": np.array([c["coef"] for c in cases]).T.copy(),
             "yaw_lo": np.array([c["yaw_lo"] for c in cases]), "yaw_hi": np.array([c["yaw_hi"] for c in cases])}
        r = twin_solve(host_twin, params, b)
        assert (r["status"] == 0).all()
        ref = np.array([c["out9"] for c in cases]).T
        assert np.max(np.abs(r["out"][6] - ref[6])) < 2e-6      # SLSQP's own accuracy bounds this comparison
        assert np.max(np.abs(r["out"][7] - ref[7])) < 2e-6
        assert np.max(np.abs(r["out"][:6] - ref[:6])) < 2e-5


def test_twin_edge_cases(pkg, host_twin, golden_dir):
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"))
    st = np.zeros((6, 4)); cf = np.zeros((5, 4))
    st[3] = [60.0, 20.0, 20.0, 20.0]          # 0: v0 above maxSpeed -> infeasible
    st[2, 1] = 0.2                             # 1: psi0 outside [-0.1, 0.1] -> infeasible
    st[4, 3] = 0.7; cf[0, 3] = 0.7             # 3: lateral offset only
    b = {"state": st, "coeffs": cf, "yaw_lo": np.full(4, -0.1), "yaw_hi": np.full(4, 0.1)}
    r = twin_solve(host_twin, params, b)
    assert list(r["status"]) == [3, 3, 0, 0]
    assert abs(r["out"][6, 2]) < 1e-9 and r["out"][7, 2] == pytest.approx(params.max_acceleration, abs=1e-6)
    assert r["out"][6, 3] > 0                  # road is to the left (cte = f(0) - y > 0): steer left
    # empty batch
    e = {"state": np.zeros((6, 0)), "coeffs": np.zeros((5, 0)), "yaw_lo": np.zeros(0), "yaw_hi": np.zeros(0)}
    assert twin_solve(host_twin, params, e)["out"].shape == (9, 0)


def test_twin_light_math(host_twin):
    """fsincos (Cody-Waite + minimax kernels) against numpy over the argument ranges the model produces."""
    import ctypes as C
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-4, 4, 20000), rng.uniform(-300, 300, 20000), rng.normal(0, 1e-3, 2000),
                        np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, 1e5 - 1, 2e5, -3e7, 9.9e8])])
    sn = np.zeros_like(x); cs = np.zeros_like(x); rc = np.zeros_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    host_twin.mpc_host_twin_math(C.c_int64(len(x)), p(x), p(sn), p(cs), p(rc))
    assert np.max(np.abs(sn - np.sin(x))) < 4e-16 and np.max(np.abs(cs - np.cos(x))) < 4e-16
    # beyond the reduction's range the evaluation is flagged, not wrong
    big = np.array([1e9, -4e12, np.inf, np.nan]); o1 = np.zeros(4); o2 = np.zeros(4); o3 = np.zeros(4)
    host_twin.mpc_host_twin_math(C.c_int64(4), p(big), p(o1), p(o2), p(o3))
    assert np.isnan(o1).all() and np.isnan(o2).all()
    # atan (road slope) and log (barrier): the solver's own kernels
    xa = np.concatenate([rng.uniform(-1, 1, 20000), rng.uniform(-60, 60, 20000), 10.0 ** rng.uniform(-12, 12, 5000),
                         np.array([0.0, 1.0, -1.0, 1.0 + 1e-16, 1e300])])
    at = np.zeros_like(xa); lg = np.zeros_like(xa)
    host_twin.mpc_host_twin_math2(C.c_int64(len(xa)), p(xa), p(at), p(lg))
    assert np.max(np.abs(at - np.arctan(xa)) / np.maximum(np.abs(np.arctan(xa)), 1e-300)) < 6e-16
    nz = xa != 0
    assert np.max(np.abs(lg[nz] - np.log(np.abs(xa[nz]))) / np.maximum(np.abs(np.log(np.abs(xa[nz]))), 1e-3)) < 5e-16


@pytest.mark.parametrize("cut", [3, 12])
def test_park_and_resume_is_bitwise_identical(pkg, host_twin, golden_dir, waypoints, cut):
    """The two-phase solve (csrc/mpc_solver.hip: park after `cut` passes, resume in another lane on another workspace
    that only receives the current iterate slot) replayed with the host build: not a bit may change."""
    import ctypes as C
    from helpers import vp
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    B = 768
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=91)
    ref = twin_solve(host_twin, params, b, want_traj=False)
    st = np.ascontiguousarray(b["state"]); cf = np.ascontiguousarray(b["coeffs"])
    yl = np.ascontiguousarray(b["yaw_lo"]); yh = np.ascontiguousarray(b["yaw_hi"])
    out = np.zeros((9, B)); status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32); parked = np.zeros(B, dtype=np.int32)
    rc = host_twin.mpc_host_twin_solve_parked(C.byref(params), C.c_int64(B), C.c_int64(B), C.c_int(cut), vp(st), vp(cf), vp(yl), vp(yh),
                                              None, vp(out), vp(status), vp(iters), vp(parked))
    assert rc == 0
    assert parked.sum() > (B // 4 if cut == 12 else B - 5)
    assert np.array_equal(out, ref["out"]) and np.array_equal(status, ref["status"]) and np.array_equal(iters, ref["iters"])


@pytest.mark.parametrize("f32", [False, True])
def test_repeated_park_and_resume_is_bitwise_identical(pkg, host_twin, golden_dir, waypoints, f32):
    """A cut schedule parks an instance several times (MpcParams.pass_cut, pass_cut_next): every 4 passes here, with
    per-instance weights (the heavy-tailed batch the schedule is made for), in both precisions -- the fp32 solver keeps one
    more value between passes (the previous output step), which has to travel too."""
    import ctypes as C
    from helpers import vp, twin_solve_f32
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    B = 512
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=92)
    w = pkg.scenarios.weight_sweep(B, params, seed=93)
    dt = np.float32 if f32 else np.float64
    if f32:
        params.precision = pkg.PRECISION_F32
        ref = twin_solve_f32(host_twin, params, b, weights=w, want_traj=False)
        fn = host_twin.mpc_host_twin_solve_reparked_f32
    else:
        ref = twin_solve(host_twin, params, b, weights=w, want_traj=False)
        fn = host_twin.mpc_host_twin_solve_reparked
    a = lambda x: np.ascontiguousarray(x, dtype=dt)
    st, cf, yl, yh, ww = a(b["state"]), a(b["coeffs"]), a(b["yaw_lo"]), a(b["yaw_hi"]), a(w)
    out = np.zeros((9, B), dtype=dt); status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32); parked = np.zeros(B, dtype=np.int32)
    rc = fn(C.byref(params), C.c_int64(B), C.c_int64(B), C.c_int(4), vp(st), vp(cf), vp(yl), vp(yh), vp(ww), vp(out), vp(status), vp(iters), vp(parked))
    assert rc == 0
    assert parked.max() >= 10 and (parked >= 2).mean() > 0.8
    assert np.array_equal(out, np.asarray(ref["out"], dtype=dt)) and np.array_equal(status, ref["status"]) and np.array_equal(iters, ref["iters"])


@pytest.mark.parametrize("f32", [False, True])
def test_garbage_inputs_get_a_status_and_stay_contained(pkg, host_twin, golden_dir, waypoints, f32):
    """NaN, infinities, 1e300, inverted bounds, negative and infinite weights (helpers.garbage_batch), one defect per
    instance: every such instance ends with a status (the solve terminates: this test finishing is the assertion), and the
    clean instances behind them are solved bit for bit as in a clean batch."""
    from helpers import garbage_batch, twin_solve_f32
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    if f32:
        params.precision = pkg.PRECISION_F32
    B = 96
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=5)
    w = pkg.scenarios.weight_sweep(B, params, seed=3)
    gb, gw, names = garbage_batch(b, w)
    solve = (lambda bb, ww: twin_solve_f32(host_twin, params, bb, weights=ww, want_traj=False)) if f32 else \
            (lambda bb, ww: twin_solve(host_twin, params, bb, weights=ww, want_traj=False))
    with np.errstate(over="ignore"):
        clean, dirty = solve(b, w), solve(gb, gw)
    n = len(names)
    assert set(np.unique(dirty["status"][:n])) <= {0, 1, 2, 3, 4}
    st = dict(zip(names, dirty["status"][:n]))
    for name in ("v0 = nan", "yaw_lo = nan", "yaw_lo > yaw_hi", "v0 = 1e6", "psi0 = 1e6"):
        assert st[name] == 3, (name, st[name])            # outside its own bounds (or no interior): rejected at set-up
    # yaw_lo == yaw_hi: IPOPT takes equal bounds as a fixed variable and solves; with its bound relaxation restated the
    # interval is 2e-8 wide and the solve goes through (psi pinned to the bound, delta = 0): a status, whichever
    assert st["yaw_lo == yaw_hi"] in (0, 2, 3)
    for name in ("c0 = nan", "w[0] = nan", "w[1] = inf", "c1 = inf", "epsi0 = -inf"):
        assert st[name] == 4, (name, st[name])            # not a number somewhere in the first evaluation
    for name in ("v0 = -5", "w[0] = -1", "all weights 0", "w[3] = 1e30"):
        assert st[name] == 0, (name, st[name])            # unusual but well-posed (a negative weight is regularised away)
    for k in ("out", "status", "iters"):
        assert np.array_equal(dirty[k][..., n:], clean[k][..., n:]), k


def test_twin_against_scipy_goldens_long_horizon_and_weights(pkg, host_twin, golden_dir):
    gold = load_golden("scipy_cross_solve_ext.json")
    for sel, over in ((lambda c: c["N"] == 25, dict(N=25, dt=0.05)), (lambda c: c["weights"] is not None, {})):
        cases = [c for c in gold["cases"] if sel(c)]
        assert len(cases) >= 6
        params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"), **over)
        b = {"state": np.array([c["state"] for c in cases]).T.copy(), "coeffs": np.array([c["coef"] for c in cases]).T.copy(),
             "yaw_lo": np.array([c["yaw_lo"] for c in cases]), "yaw_hi": np.array([c["yaw_hi"] for c in cases])}
        w = np.array([c["weights"] for c in cases]).T.copy() if cases[0]["weights"] is not None else None
        r = twin_solve(host_twin, params, b, weights=w)
        assert (r["status"] == 0).all()
        ref = np.array([c["out9"] for c in cases]).T
        assert np.max(np.abs(r["out"][6] - ref[6])) < 5e-6
        assert np.max(np.abs(r["out"][7] - ref[7])) < 5e-6
        assert np.max(np.abs(r["out"][:6] - ref[:6])) < 5e-5


def test_acceptable_level_termination_matches_oracle(pkg, host_twin, golden_dir, waypoints):
    """IPOPT's acceptable-level termination (MpcParams.acceptable_*; OrcSolveOptions likewise).  With IPOPT's defaults no instance of
    these batches ends "acceptable" (they converge): the statuses are what they were.  With the test's own settings -- an acceptable
    level of 1e-3 that most instances pass two iterations before they converge, two acceptable iterates in a row -- those instances
    must end with MPC_STATUS_ACCEPTABLE = 6 in the device solver (CPU build) AND in the oracle: the same instances, after the same
    number of iterations, at the same point (an iterate a step or two short of the solution); and with acceptable_iter = 0 the
    machinery is off."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    B = 48
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=31)
    r0 = twin_solve(host_twin, params, b)
    assert (r0["status"] == 0).all()
    q = params.copy(); q.acceptable_tol = 1e-3; q.acceptable_iter = 2
    r = twin_solve(host_twin, q, b)
    acc = r["status"] == 6
    assert acc.sum() >= 36 and set(np.unique(r["status"])) <= {0, 6}, np.bincount(r["status"], minlength=7)     # (a few converge outright)
    assert (r["iters"][acc] < r0["iters"][acc]).all() and np.array_equal(r["iters"][~acc], r0["iters"][~acc])
    cfg = O.load_config("config-fast.json")
    ref = oracle_solve_batch(cfg, b, range(B), opt=O.default_options(acceptable_tol=1e-3, acceptable_iter=2))
    assert np.array_equal(r["status"], ref["status"]), (np.bincount(r["status"], minlength=7), np.bincount(ref["status"], minlength=7))
    assert np.array_equal(r["iters"], ref["iters"])
    assert_parity(r["out"], ref["out"], r["traj"], ref["traj"], "acceptable level")
    assert np.max(np.abs(r["out"][6] - r0["out"][6])) < 1e-3              # (an acceptable iterate, not the solution)
    off = q.copy(); off.acceptable_iter = 0
    r2 = twin_solve(host_twin, off, b)
    assert np.array_equal(r2["status"], r0["status"]) and np.array_equal(r2["iters"], r0["iters"]) and np.array_equal(r2["out"], r0["out"])
    # the unscaled tests of the convergence check: a constraint-violation tolerance next to nothing can meet (exactly zero residuals:
    # an instance or two) turns the successes into the cap
    strict = params.copy(); strict.constr_viol_tol = 0.0; strict.acceptable_iter = 0; strict.max_iter = 40
    r3 = twin_solve(host_twin, strict, b)
    assert (r3["status"] == 1).sum() >= B - 4
