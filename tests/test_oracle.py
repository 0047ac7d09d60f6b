"""The oracle against every known answer that exists for this path (CPU only).

There are no reference unit tests.  What pins the oracle:
  * the values recorded in BASELINE.md section 2 for the reference's own demo scenario (src/test.cpp:45-50);
  * an independent numpy/scipy-SLSQP cross-solve (tests/golden/scipy_cross_solve.json);
  * curves digitised from the reference's own IPOPT result figure (tests/golden/plot_anchors_10-01-2.json);
  * finite differences of the restated FG_eval for the hand-derived derivatives;
  * a solver-independent KKT certificate.
"""
import numpy as np
import pytest

import oracle_lib as O
from helpers import TEST_CPP, load_golden


def _xi(cfg, state):
    N = cfg.N
    xi = np.zeros(O.nvars(N))
    for k in range(6):
        xi[k * N] = state[k]
    return xi


def test_config_load_matches_baseline_md():
    # BASELINE.md / SURVEY.md section 8c, "Values captured": Config::load derived values
    cfg = O.load_config("config-stable.json")
    assert cfg.N == 10 and cfg.dt == 0.1
    assert cfg.max_steering == pytest.approx(0.43633231299858238, abs=1e-16)
    assert cfg.max_acceleration == pytest.approx(4.4703888888888885, abs=1e-15)
    assert cfg.max_deceleration == pytest.approx(-8.940777777777777, abs=1e-15)
    assert cfg.max_speed == pytest.approx(53.644666666666666, abs=1e-13)
    assert cfg.steer_speeds[0] == pytest.approx(53.644666666666666, abs=1e-10)   # 100 mph x speedScale 1.2
    assert cfg.lookahead == pytest.approx(0.1)
    assert cfg.n_weights == 12 and cfg.n_steers == 9 and cfg.n_yaw_changes == 10 and cfg.n_yaw_change_speeds == 11
    fast = O.load_config("config-fast.json")
    assert fast.max_speed == pytest.approx(145 * 1609.34 / 3600.0)
    assert fast.steer_adj_thresh == 0.6 and fast.steer_adj_ratio == 0.021
    nolat = O.load_config("config-no-latency.json")
    assert nolat.latency == 0 and nolat.lookahead == 0.0 and nolat.steer_adj_ratio == 0.01


def test_run_preprocessing_known_answers():
    # src/test.cpp:45-50 through MPC::run()'s pre-solve half; expected values from BASELINE.md section 2
    cfg = O.load_config("config-stable.json")
    pre, px, py = O.run_pre(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    assert (px[0], py[0]) == (pytest.approx(-3.122981, abs=1e-6), pytest.approx(-0.140215, abs=1e-6))
    assert (px[-1], py[-1]) == (pytest.approx(82.122303, abs=1e-6), pytest.approx(18.276468, abs=1e-6))
    assert pre.nc == 3                                   # adaptive fit stops at degree 2
    assert list(pre.coef)[:3] == pytest.approx([-0.176556100493, -0.0259923242089, 0.00302916583378], abs=1e-12)
    assert pre.state[4] == pytest.approx(-0.176556100493, abs=1e-12)
    assert pre.state[5] == pytest.approx(0.0259864731012, abs=1e-12)
    assert pre.max_yaw_change == pytest.approx(0.484345392317, abs=1e-11)
    assert (pre.yaw_low, pre.yaw_high) == (-0.1, pytest.approx(0.484345392317, abs=1e-11))
    assert pre.max_speed == pytest.approx(53.6446666667, abs=1e-9)
    assert pre.target_speed == pytest.approx(53.6446666667, abs=1e-9)


def test_fg_eval_cost_at_start_point():
    # reference FG_eval cost at xi recorded in BASELINE.md section 2: 26626.9119489
    cfg = O.load_config("config-stable.json")
    pre, _, _ = O.run_pre(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    xi = _xi(cfg, list(pre.state))
    fg = O.fg_eval(cfg, list(pre.coef)[:pre.nc], xi, xi)
    assert fg[0] == pytest.approx(26626.9119489, abs=1e-6)
    # initial-state residual rows equal the state (MPC.cpp:116-121); v row of stage 1 is 0 - v0
    assert fg[1 + 3 * cfg.N] == pytest.approx(26.6806)
    assert fg[1 + 3 * cfg.N + 1] == pytest.approx(-26.6806)


def test_solve_known_answer_test_cpp():
    # cross-solved values recorded in BASELINE.md section 2 (scipy SLSQP and trust-constr agreed to ~2e-9)
    cfg = O.load_config("config-stable.json")
    pre, _, _ = O.run_pre(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    cfg.yaw_low, cfg.yaw_high = pre.yaw_low, pre.yaw_high
    st, o9, tx, ty, info = O.mpc_solve(cfg, list(pre.state), list(pre.coef)[:pre.nc])
    assert st == 0 and info.kkt_error <= 1e-8
    assert o9[6] == pytest.approx(0.0024133755, abs=5e-9)       # delta0
    assert o9[7] == pytest.approx(4.47038889, abs=1e-7)         # a0 on its upper bound
    assert o9[2] == pytest.approx(0.0024116, abs=1e-7)          # psi1
    assert o9[4] == pytest.approx(-0.1072304343, abs=1e-8)      # cte1
    assert o9[5] == pytest.approx(0.028398096, abs=1e-8)        # epsi1
    assert o9[8] == pytest.approx(6243.44368073, abs=1e-5)      # J*
    assert tx[0] == 0.0 and tx[1] == pytest.approx(2.66806, abs=1e-9)
    # F3: the oracle's LIVE mode (re-decide the branches at the solution and re-solve until they stop
    # changing) is NOT the reference semantics; it is kept for comparison only.  Here it switches on the
    # w[6] a^2 term (a > 0 at the solution) and moves delta0 by < 1e-5.
    live = O.default_options(branch_mode=1)
    stl, o9l, _, _, _ = O.mpc_solve(cfg, list(pre.state), list(pre.coef)[:pre.nc], live)
    assert stl == 0 and abs(o9l[6] - o9[6]) < 1e-5 and o9l[8] != o9[8]


def test_solver_options_do_not_move_the_solution():
    cfg = O.load_config("config-stable.json")
    pre, _, _ = O.run_pre(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    cfg.yaw_low, cfg.yaw_high = pre.yaw_low, pre.yaw_high
    ref = O.mpc_solve(cfg, list(pre.state), list(pre.coef)[:pre.nc])[1]
    for kw in (dict(lam_init_ls=0), dict(obj_scaling=0), dict(lam_init_ls=0, obj_scaling=0)):
        st, o9, _, _, _ = O.mpc_solve(cfg, list(pre.state), list(pre.coef)[:pre.nc], O.default_options(**kw))
        assert st == 0
        assert np.max(np.abs(o9[:8] - ref[:8])) < 1e-7


def test_against_scipy_cross_solve():
    gold = load_golden("scipy_cross_solve.json")
    worst = 0.0
    for cs in gold["cases"]:
        cfg = O.load_config(cs["config"])
        cfg.yaw_low, cfg.yaw_high = cs["yaw_lo"], cs["yaw_hi"]
        xi = _xi(cfg, cs["state"])
        assert O.fg_eval(cfg, cs["coef"], xi, xi)[0] == pytest.approx(cs["cost_at_xi"], rel=1e-12)
        st, o9, _, _, info = O.mpc_solve(cfg, cs["state"], cs["coef"])
        assert st == 0, cs["name"]
        ref = np.array(cs["out9"])
        # SLSQP's own accuracy is ~1e-7 (it stops on objective change); the oracle must not cost more,
        # up to the interior-point offset of the active bounds (~ n_active * mu / obj_scaling)
        assert o9[8] <= ref[8] * (1 + 2e-8) + 1e-7, cs["name"]
        assert abs(o9[6] - ref[6]) < 2e-6 and abs(o9[7] - ref[7]) < 2e-6, (cs["name"], o9[6], ref[6])
        assert np.max(np.abs(o9[:6] - ref[:6])) < 2e-5, cs["name"]
        worst = max(worst, abs(o9[6] - ref[6]))
    assert worst < 2e-6


def test_closed_loop_matches_reference_figure():
    """The 26-sample closed loop of src/test.cpp:67-111 against the curves digitised from the reference's
    IPOPT figure examples/10-01-2.png (plot precision: ~1.5 px)."""
    anchors = load_golden("plot_anchors_10-01-2.json")
    cfg = O.load_config("config-stable.json")
    st, out8, _, _, pre, _ = O.mpc_run(cfg, TEST_CPP["pose"], TEST_CPP["ptsx"], TEST_CPP["ptsy"])
    assert st == 0
    coef = list(pre.coef)[:pre.nc]
    cte, epsi, delta, v = [out8[6]], [out8[7]], [out8[4] * cfg.max_steering], [out8[3]]
    state = [out8[0], out8[1], out8[2], out8[3], out8[6], out8[7]]          # test.cpp:79-80
    for _ in range(25):
        st, o9, _, _, _ = O.mpc_solve(cfg, state, coef)                     # test.cpp:85 (cold start each time)
        assert st == 0
        cte.append(o9[4]); epsi.append(o9[5]); delta.append(o9[6]); v.append(o9[3])
        state = list(o9[:6])                                                # test.cpp:97
    px = anchors["pixel_value"]
    a = anchors["curves"]
    assert np.max(np.abs(np.array(cte) - a["cte"])) <= 2.5 * px["cte"]
    assert np.max(np.abs(np.array(epsi)[5:] - a["epsi"][5:])) <= 3.0 * px["epsi"]
    assert np.max(np.abs(np.array(epsi)[:5] - a["epsi"][:5])) <= 5.0 * px["epsi"]   # steep first segments
    assert np.max(np.abs(np.array(delta)[1:] - a["delta"][1:])) <= 2.5 * px["delta"]  # delta[0]: see fixture script
    assert np.max(np.abs(np.array(v) - a["v"])) <= 2.0 * px["v"]
    # v rises by exactly maxAcceleration*dt per step: all accelerations sit on their bound
    assert np.allclose(np.diff(v), cfg.max_acceleration * cfg.dt, atol=1e-6)


def test_derivatives_match_finite_differences():
    rng = np.random.default_rng(5)
    cfg = O.load_config("config-fast.json")
    N = cfg.N
    n, m = O.nvars(N), 6 * N
    coef = [0.4, -0.05, 0.004, -2e-5, 1e-7]
    z = rng.normal(size=n) * 0.3
    z[3 * N:4 * N] += 20.0                               # speeds
    xi = _xi(cfg, [0, 0, 0, 20.0, 0.4, 0.05])
    g, J = O.fg_grad(cfg, coef, z, xi)
    lam = rng.normal(size=m)
    H = O.lag_hess(cfg, coef, z, 0.7, lam, xi)
    h = 1e-6
    gn = np.zeros(n); Jn = np.zeros((m, n))
    for i in range(n):
        e = np.zeros(n); e[i] = h
        fp, fm = O.fg_eval(cfg, coef, z + e, xi), O.fg_eval(cfg, coef, z - e, xi)
        gn[i] = (fp[0] - fm[0]) / (2 * h)
        Jn[:, i] = (fp[1:] - fm[1:]) / (2 * h)
    assert np.max(np.abs(g - gn)) < 1e-4 * max(1, np.max(np.abs(g)))
    assert np.max(np.abs(J - Jn)) < 1e-6
    # Hessian of the Lagrangian by differencing the analytic gradient of L
    Hn = np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n); e[i] = h
        gp, Jp = O.fg_grad(cfg, coef, z + e, xi)
        gm, Jm = O.fg_grad(cfg, coef, z - e, xi)
        Hn[:, i] = (0.7 * (gp - gm) + (Jp - Jm).T @ lam) / (2 * h)
    assert np.max(np.abs(H - H.T)) < 1e-12
    assert np.max(np.abs(H - Hn)) < 1e-5 * max(1, np.max(np.abs(H)))


def test_kkt_certificate_of_oracle_solutions(pkg, waypoints, golden_dir):
    import os
    params = pkg.params_from_json(os.path.join(golden_dir, "config-fast.json"))
    b = pkg.scenarios.lake_track_batch(32, params, waypoints, seed=11)
    cfg = O.load_config("config-fast.json")
    for i in range(32):
        cfg.yaw_low, cfg.yaw_high = float(b["yaw_lo"][i]), float(b["yaw_hi"][i])
        st, o9, _, _, info, sol = O.mpc_solve(cfg, b["state"][:, i], b["coeffs"][:, i], want_sol=True)
        assert st == 0
        tot, stat, prim, bnd = O.kkt_certificate(cfg, b["state"][:, i], b["coeffs"][:, i], sol, active_tol=1e-6)
        # the returned point is IPOPT's: solved inside bounds relaxed by 1e-8 max(1, |b|), then projected into the caller's
        # (honor_original_bounds): an acceleration that sat on its relaxed bound or a heading on its bound leave up to a few 1e-8 of dynamics residual
        assert prim < 1e-7 and bnd < 1e-9 and stat < 1e-6, (i, stat, prim, bnd)
        # a perturbed point must fail the certificate
        bad = sol.copy(); bad[6 * cfg.N] += 1e-3
        assert O.kkt_certificate(cfg, b["state"][:, i], b["coeffs"][:, i], bad, active_tol=1e-6)[0] > 1e-5


def test_edge_cases():
    cfg = O.load_config("config-stable.json")
    cfg.yaw_low, cfg.yaw_high = -0.1, 0.1
    # initial speed above maxSpeed / psi outside its bounds: the reference NLP is infeasible -> flagged
    assert O.mpc_solve(cfg, [0, 0, 0, 60.0, 0, 0], [0, 0, 0, 0, 0])[0] == 3
    assert O.mpc_solve(cfg, [0, 0, 0.2, 20.0, 0, 0], [0, 0, 0, 0, 0])[0] == 3
    # already on the reference line, at the speed target: no steering, zero cost apart from nothing to gain
    st, o9, _, _, _ = O.mpc_solve(cfg, [0, 0, 0, 20.0, 0, 0], [0, 0, 0, 0, 0])
    assert st == 0 and abs(o9[6]) < 1e-9 and o9[7] == pytest.approx(cfg.max_acceleration, abs=1e-6)
    # mirror symmetry: flipping the road polynomial and the errors flips the steering
    a = O.mpc_solve(cfg, [0, 0, 0, 25.0, 0.7, -0.05], [0.7, 0.05, 0.001, 0, 0])[1]
    b = O.mpc_solve(cfg, [0, 0, 0, 25.0, -0.7, 0.05], [-0.7, -0.05, -0.001, 0, 0])[1]
    assert a[6] == pytest.approx(-b[6], abs=1e-8) and a[8] == pytest.approx(b[8], rel=1e-9)
    # polyfit: exact recovery of a cubic through 6 points, adaptive order picks 3 coefficients for a parabola
    x = np.array([-3.0, 10, 25, 40, 60, 80.0])
    y = 0.3 - 0.02 * x + 0.003 * x * x
    coef = np.zeros(8)
    nc = O.lib().orc_road_fit(O.dptr(x), O.dptr(y), 6, 5, 0.5, O.dptr(coef), None)
    assert nc == 3 and coef[:3] == pytest.approx([0.3, -0.02, 0.003], abs=1e-10)
    # throttle map, Vehicle.cpp:81-103
    L = O.lib()
    assert L.orc_compute_throttle(cfg, 0.0005, 26.8, cfg.max_acceleration, cfg.max_deceleration) == pytest.approx(26.8 / cfg.max_speed)
    assert L.orc_compute_throttle(cfg, -20.0, 26.8, cfg.max_acceleration, cfg.max_deceleration) == -1
    assert L.orc_normalize_angle(4.0) == pytest.approx(4.0 - 2 * np.pi)


def test_against_scipy_cross_solve_long_horizon_and_weights():
    """BASELINE.json configs[3] (N=25, dt=0.05) and configs[4] (per-instance weights) against the independent SLSQP
    cross-solve (tests/golden/scipy_cross_solve_ext.json, generator make_golden_ext.py)."""
    gold = load_golden("scipy_cross_solve_ext.json")
    assert len(gold["cases"]) == 16
    for cs in gold["cases"]:
        cfg = O.load_config(cs["config"], N=cs["N"], dt=cs["dt"])
        if cs["weights"] is not None:
            for q in range(12):
                cfg.weights[q] = cs["weights"][q]
        cfg.yaw_low, cfg.yaw_high = cs["yaw_lo"], cs["yaw_hi"]
        xi = _xi(cfg, cs["state"])
        assert O.fg_eval(cfg, cs["coef"], xi, xi)[0] == pytest.approx(cs["cost_at_xi"], rel=1e-12)
        st, o9, _, _, info = O.mpc_solve(cfg, cs["state"], cs["coef"])
        assert st == 0, cs["name"]
        ref = np.array(cs["out9"])
        assert o9[8] <= ref[8] * (1 + 2e-8) + 1e-6, cs["name"]
        # a0 is weakly determined in the interior: every case here has it on its bound
        assert abs(o9[6] - ref[6]) < 5e-6 and abs(o9[7] - ref[7]) < 5e-6, (cs["name"], o9[6], ref[6])
        assert np.max(np.abs(o9[:6] - ref[:6])) < 5e-5, cs["name"]


def test_second_order_correction_is_inert_on_well_posed_instances():
    """OrcSolveOptions.max_soc (default 0: the second-order correction is NOT part of the restated algorithm; DESIGN.md section 3 has
    what it does to the hard instances of SURVEY's population).  On the well-posed instances the reference itself holds -- the
    test.cpp scenario and its commented snapshots -- first trial points are accepted, so IPOPT's max_soc = 4 changes neither
    status nor iteration count nor point."""
    from helpers import TEST_CPP_COMMENTED
    assert O.default_options().max_soc == 0
    cfg = O.load_config("config-stable.json")
    for sc in [TEST_CPP] + list(TEST_CPP_COMMENTED):
        pre, _, _ = O.run_pre(cfg, sc["pose"], sc["ptsx"], sc["ptsy"])
        cfg.yaw_low, cfg.yaw_high = pre.yaw_low, pre.yaw_high
        res = {}
        for soc in (0, 4):
            st, o9, tx, ty, info = O.mpc_solve(cfg, list(pre.state), list(pre.coef)[:pre.nc], O.default_options(max_soc=soc))
            res[soc] = (st, np.array(o9), info.iterations, info.n_soc_accepted)
        assert res[0][0] == res[4][0] == 0 and res[4][3] == 0 and res[0][2] == res[4][2]
        assert np.max(np.abs(res[0][1] - res[4][1])) < 1e-12
