"""Shared test helpers (test infrastructure)."""
import ctypes as C
import json
import os

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the reference's own demo scenario, src/test.cpp:45-50
TEST_CPP = dict(
    ptsx=[-145.1165, -158.3417, -164.3164, -169.3365, -175.4917, -176.9617],
    ptsy=[4.339378, -17.42898, -30.18062, -42.84062, -66.52898, -76.85062],
    pose=[-146.7283, 1.660802, 4.125825, 26.6806, 0.0, 0.0])

# the four telemetry snapshots that sit commented out in src/test.cpp:18-43 (inputs only: the reference holds no outputs
# for them).  The third one has no waypoint lists of its own in the file; it is given the first list (:18-19), which is
# the stretch of track its position lies on.
_WP_A = ([-134.97, -145.1165, -158.3417, -164.3164, -169.3365, -175.4917], [18.404, 4.339378, -17.42898, -30.18062, -42.84062, -66.52898])
_WP_B = ([-164.3164, -169.3365, -175.4917, -176.9617, -176.8864, -175.0817], [-30.18062, -42.84062, -66.52898, -76.85062, -90.64063, -100.3206])
_WP_D = ([-61.09, -78.29172, -93.05002, -107.7717, -123.3917, -134.97], [92.88499, 78.73102, 65.34102, 50.57938, 33.37102, 18.404])
TEST_CPP_COMMENTED = [
    dict(ptsx=_WP_A[0], ptsy=_WP_A[1], pose=[-146.8912, 2.129487, 0.4009452, 13.87815, 0.0, 0.0]),      # test.cpp:18-23
    dict(ptsx=_WP_B[0], ptsy=_WP_B[1], pose=[-166.0726, -29.59644, 4.088, 30.62756, 0.0, 0.0]),          # test.cpp:25-31
    dict(ptsx=_WP_A[0], ptsy=_WP_A[1], pose=[-144.7913, 3.767814, 0.03732295, 10.32361, 0.0, 0.0]),      # test.cpp:33-36
    dict(ptsx=_WP_D[0], ptsy=_WP_D[1], pose=[-61.97283, 93.53992, 3.857562, 33.06046, 0.0, 0.0]),        # test.cpp:38-43
]

# Instances that leave the central path: the UNFILTERED lake-track draw (seed 45, 16 384 instances,
# config-fast.json) contains the situations the generator normally rejects -- waypoint windows that double back, so that
# the "road" is a degree-4 fit with a cte of hundreds of metres.  On those the line search runs out of step length, i.e.
# where IPOPT would enter its restoration phase (not restated; stand-in: one restart with zero multipliers), or the
# iteration cap strikes.  index -> (status, iterations) that oracle and device solver BOTH report, with the same point:
OFF_PATH_BATCH = dict(config="config-fast.json", B=16384, seed=45)
# (round 3: with IPOPT's bound_relax_factor restated in both solvers three of the counts moved -- 2080: 43 -> 42, 8676: 129 -> 71,
# 12533: 376 -> 375 -- in both solvers alike)
OFF_PATH_INSTANCES = {235: (2, 47), 2080: (2, 42), 11515: (2, 8), 12533: (1, 375),          # restoration stand-in fails / cap
                      1340: (0, 198), 8676: (0, 71), 9386: (0, 124), 9966: (0, 110), 10296: (0, 105)}   # converge after 70-200 iterations
# 6049 (cte0 = -1143 m) and 1473 converge in both solvers, to DIFFERENT local minima (6049: delta0 -0.436 vs +0.042): not parity cases.

# The same at the long horizon: configs[3]'s full batch (262 144 lake-track instances, N = 25, dt = 0.05, PRNG stream 3) has
# four instances on which the line search runs out of step length in BOTH solvers (same iteration, same point), and one
# (the last row) that sits on the rounding floor of the tolerance: it converges, and only the polish could still lose it.
OFF_PATH_N25 = dict(config="config-stable.json", N=25, dt=0.05, expect=[(2, 69), (2, 72), (2, 55), (2, 48), (0, None)], rows=np.array([
    [0.0, 0.0, 0.0, 42.037103799678775, -0.7602803429124025, 0.10866518072779344, -0.7602803429124025, -0.10909492116528459, 0.011820842324803153, -0.0018921443497710185, 0.0, -1.2926142189803314, 0.1],
    [0.0, 0.0, 0.0, 39.67702028664985, -0.4886522467329021, 0.10626125436148554, -0.4886522467329021, -0.10666301708143032, 0.01213399309381295, -0.0018268937770139536, 0.0, -1.2766359600284345, 0.1],
    [0.0, 0.0, 0.0, 39.036764543866774, 0.3456441543053598, 0.12874741007748614, 0.3456441543053598, -0.12946352643669048, 0.015420947915147526, -0.001961261791901044, 0.0, -1.2506188534670613, 0.1],
    [0.0, 0.0, 0.0, 51.24670367204065, -1.026025721766231, 0.0933026209449351, -1.026025721766231, -0.09357431194811669, 0.0038347925967602116, -0.001994599149009969, 0.0, -1.418708227207186, 0.1],
    [0.0, 0.0, 0.0, 52.82001239973368, -0.7920241929084179, -0.035491291798923075, -0.7920241929084179, 0.035506201297674365, 0.005042799086039826, 5.937535751267709e-05, 0.0, -0.1, 0.8422130108717072],
]))   # columns: state[6], coeffs[5], yaw_lo, yaw_hi

# stated fp64 tolerances (SURVEY.md section 8d / BASELINE.md section 4)
TOL_STEER = 1e-6   # rad, delta0
TOL_ACCEL = 1e-6   # m/s^2, a0
TOL_TRAJ = 1e-5    # m, predicted trajectory points / step-1 state
TOL_COST_REL = 1e-7
# Stated tolerances of the MPC_PRECISION_F32 mode against the fp64 oracle (BASELINE.json configs[4]; SURVEY.md section 8d expects
# ~1e-3 rad), written down BEFORE the first measurement of round 3: every instance of a batch, velocity weight 0 included, long
# horizons included.  The mode runs the interior-point iteration in fp32 down to the barrier parameter MpcParams.mixed_switch_mu
# and finishes every instance in fp64 (f32_finish = 1, the default): what remains is the rounding of the fp32 inputs and outputs.
F32_TOL_STEER = 1e-3     # rad
F32_TOL_ACCEL = 1e-3     # m/s^2
F32_TOL_STATE = 1e-3     # m / rad / m/s, step-1 state
F32_TOL_TRAJ = 1e-2      # m, predicted trajectory
F32_TOL_COST_REL = 1e-5
# The pure fp32 solver (f32_finish = 0; round 2's mode) cannot meet IPOPT's tolerance: it stops at tol_f32 = 5e-4 with a barrier
# floor of 2e-5, and its tolerances were set after measuring (65 536-instance soak: steer max 2.4e-3, a0 max 6.8e-2, state max
# 6.8e-3, trajectory max 0.12 m); velocity weight 0 is outside its specification (it may pick the other bang of a bang-bang a0).
F32PURE_TOL_STEER = 5e-3
F32PURE_TOL_ACCEL = 1e-1
F32PURE_TOL_STATE = 1e-2
F32PURE_TOL_TRAJ = 0.3
F32PURE_TOL_COST_REL = 1e-4


def vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def twin_solve(twin, params, batch, weights=None, want_traj=True):
    """Run the TEST-ONLY host twin on a batch dict (state, coeffs, yaw_lo, yaw_hi)."""
    st = np.ascontiguousarray(batch["state"], dtype=np.float64)
    cf = np.ascontiguousarray(batch["coeffs"], dtype=np.float64)
    yl = np.ascontiguousarray(batch["yaw_lo"], dtype=np.float64)
    yh = np.ascontiguousarray(batch["yaw_hi"], dtype=np.float64)
    B = st.shape[1]
    out = np.zeros((9, B)); traj = np.zeros((2 * params.N, B)) if want_traj else None
    status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
    w = np.ascontiguousarray(weights, dtype=np.float64) if weights is not None else None
    rc = twin.mpc_host_twin_solve(C.byref(params), C.c_int64(B), C.c_int64(B), vp(st), vp(cf), vp(yl), vp(yh), vp(w),
                                  vp(out), vp(traj), vp(status), vp(iters))
    assert rc == 0
    return {"out": out, "traj": traj, "status": status, "iters": iters}


def twin_solve_f32(twin, params, batch, weights=None, want_traj=True):
    """The MPC_PRECISION_F32 solver of the device header, built for the CPU (test-only host twin): float32 in and out."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    st, cf, yl, yh = f(batch["state"]), f(batch["coeffs"]), f(batch["yaw_lo"]), f(batch["yaw_hi"])
    B = st.shape[1]
    out = np.zeros((9, B), np.float32); traj = np.zeros((2 * params.N, B), np.float32) if want_traj else None
    status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
    w = f(weights) if weights is not None else None
    rc = twin.mpc_host_twin_solve_f32(C.byref(params), C.c_int64(B), C.c_int64(B), vp(st), vp(cf), vp(yl), vp(yh), vp(w),
                                      vp(out), vp(traj), vp(status), vp(iters))
    assert rc == 0
    return {"out": out, "traj": traj, "status": status, "iters": iters}


def twin_solve_mixed(twin, params, batch, weights=None, want_traj=True):
    """MPC_PRECISION_F32 as shipped (f32_finish = 1), replayed by the test-only host build: the fp32 solver up to MPC_PROMOTE, then
    the fp64 solver on the converted iterate; float32 in and out.  Also returns the iterations of the fp32 phase."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    st, cf, yl, yh = f(batch["state"]), f(batch["coeffs"]), f(batch["yaw_lo"]), f(batch["yaw_hi"])
    B = st.shape[1]
    out = np.zeros((9, B), np.float32); traj = np.zeros((2 * params.N, B), np.float32) if want_traj else None
    status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32); it32 = np.zeros(B, dtype=np.int32)
    w = f(weights) if weights is not None else None
    rc = twin.mpc_host_twin_solve_mixed(C.byref(params), C.c_int64(B), C.c_int64(B), vp(st), vp(cf), vp(yl), vp(yh), vp(w),
                                        vp(out), vp(traj), vp(status), vp(iters), vp(it32))
    assert rc == 0
    return {"out": out, "traj": traj, "status": status, "iters": iters, "iters_f32": it32}


def twin_solve_mixed_f64(twin, params, batch, weights=None, want_traj=True):
    """An fp64 handle with f64_f32_start in effect, replayed by the test-only host build (float64 in and out)."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    st, cf, yl, yh = f(batch["state"]), f(batch["coeffs"]), f(batch["yaw_lo"]), f(batch["yaw_hi"])
    B = st.shape[1]
    out = np.zeros((9, B)); traj = np.zeros((2 * params.N, B)) if want_traj else None
    status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32); it32 = np.zeros(B, dtype=np.int32)
    w = f(weights) if weights is not None else None
    rc = twin.mpc_host_twin_solve_mixed_f64(C.byref(params), C.c_int64(B), C.c_int64(B), vp(st), vp(cf), vp(yl), vp(yh), vp(w),
                                            vp(out), vp(traj), vp(status), vp(iters), vp(it32))
    assert rc == 0
    return {"out": out, "traj": traj, "status": status, "iters": iters, "iters_f32": it32}


def oracle_solve_batch(cfg, batch, idx, opt=None, weights=None):
    """Oracle MPC::solve for the selected instances -> dict of arrays (out [9,n], traj [2N,n], status, iters)."""
    n = len(idx)
    out = np.zeros((9, n)); traj = np.zeros((2 * cfg.N, n)); status = np.zeros(n, dtype=np.int32)
    iters = np.zeros(n, dtype=np.int32)
    for j, i in enumerate(idx):
        cfg.yaw_low = float(batch["yaw_lo"][i]); cfg.yaw_high = float(batch["yaw_hi"][i])
        if weights is not None:
            for q in range(12):
                cfg.weights[q] = float(weights[q, i])
        st, o9, tx, ty, info = O.mpc_solve(cfg, batch["state"][:, i], batch["coeffs"][:, i], opt)
        out[:, j] = o9; traj[:cfg.N, j] = tx; traj[cfg.N:, j] = ty; status[j] = st; iters[j] = info.iterations
    return {"out": out, "traj": traj, "status": status, "iters": iters}


def assert_parity(got_out, ref_out, got_traj=None, ref_traj=None, what="", tol_accel=TOL_ACCEL):
    d_state = np.max(np.abs(got_out[:6] - ref_out[:6]))
    d_steer = np.max(np.abs(got_out[6] - ref_out[6]))
    d_acc = np.max(np.abs(got_out[7] - ref_out[7]))
    d_cost = np.max(np.abs(got_out[8] - ref_out[8]) / np.maximum(1.0, np.abs(ref_out[8])))
    assert d_steer <= TOL_STEER, "%s max |d steer| = %g rad" % (what, d_steer)
    assert d_acc <= tol_accel, "%s max |d accel| = %g" % (what, d_acc)
    assert d_state <= TOL_TRAJ, "%s max |d step-1 state| = %g" % (what, d_state)
    assert d_cost <= TOL_COST_REL, "%s max rel |d cost| = %g" % (what, d_cost)
    if got_traj is not None and ref_traj is not None:
        d_t = np.max(np.abs(got_traj - ref_traj))
        assert d_t <= TOL_TRAJ, "%s max |d trajectory| = %g m" % (what, d_t)
    return d_steer, d_acc, d_state


def load_golden(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", name)))


def garbage_batch(batch, weights):
    """Copies of (batch, weights) whose first instances carry inputs no caller should send -- NaN, infinities, 1e300,
    inverted or empty yaw bounds, negative / infinite / all-zero weights -- one defect per instance.  Returns
    (batch, weights, names).  What the solver owes such an instance: a status, no hang, and no effect on its neighbours."""
    b = {k: np.array(batch[k], dtype=np.float64, copy=True) for k in ("state", "coeffs", "yaw_lo", "yaw_hi")}
    w = np.array(weights, dtype=np.float64, copy=True)
    defects = [
        ("v0 = nan", lambda i: b["state"].__setitem__((3, i), np.nan)),
        ("cte0 = inf", lambda i: b["state"].__setitem__((4, i), np.inf)),
        ("epsi0 = -inf", lambda i: b["state"].__setitem__((5, i), -np.inf)),
        ("x0 = 1e300", lambda i: b["state"].__setitem__((0, i), 1e300)),
        ("c0 = nan", lambda i: b["coeffs"].__setitem__((0, i), np.nan)),
        ("c3 = 1e200", lambda i: b["coeffs"].__setitem__((3, i), 1e200)),
        ("c1 = inf", lambda i: b["coeffs"].__setitem__((1, i), np.inf)),
        ("yaw_lo = nan", lambda i: b["yaw_lo"].__setitem__(i, np.nan)),
        ("yaw_hi = inf", lambda i: b["yaw_hi"].__setitem__(i, np.inf)),
        ("yaw_lo = -inf", lambda i: b["yaw_lo"].__setitem__(i, -np.inf)),
        ("yaw_lo > yaw_hi", lambda i: (b["yaw_lo"].__setitem__(i, 1.0), b["yaw_hi"].__setitem__(i, -1.0))),
        ("yaw_lo == yaw_hi", lambda i: (b["yaw_lo"].__setitem__(i, 0.0), b["yaw_hi"].__setitem__(i, 0.0))),
        ("v0 = 1e6", lambda i: b["state"].__setitem__((3, i), 1e6)),
        ("v0 = -5", lambda i: b["state"].__setitem__((3, i), -5.0)),
        ("psi0 = 1e6", lambda i: b["state"].__setitem__((2, i), 1e6)),
        ("w[0] = nan", lambda i: w.__setitem__((0, i), np.nan)),
        ("w[0] = -1", lambda i: w.__setitem__((0, i), -1.0)),
        ("all weights 0", lambda i: w.__setitem__((slice(None), i), 0.0)),
        ("w[1] = inf", lambda i: w.__setitem__((1, i), np.inf)),
        ("w[3] = 1e30", lambda i: w.__setitem__((3, i), 1e30)),
    ]
    for i, (_, fn) in enumerate(defects):
        fn(i)
    return b, w, [d[0] for d in defects]


def f32_forks(got, ref, ok):
    """Instances on which an MPC_PRECISION_F32 solve (fp32 iterations, fp64 finish) and the fp64 reference both converge, to
    DIFFERENT local minima: beyond the stated tolerances in delta0 / a0 / step-1 state / trajectory while the costs agree to 1e-3
    relative.  The NLP is non-convex and on a flat objective (velocity weight 0, steering weight 1) the fp32 iterations can lead
    into a neighbouring basin; measured on configs[4]'s 131 072-instance share: one such instance (the mixed solve found the lower
    cost) by the first step; with the trajectory in the comparison, on a 65 536-instance soak: three (first steps equal to 1e-4,
    far ends of the trajectories 0.01-0.67 m apart, costs within 3e-5).  The tests bound them at one in 10 000.
    Returns (fork mask, wrong mask): `wrong` = beyond tolerance with costs that do NOT agree."""
    g, r = got["out"].astype(np.float64), ref["out"]
    far = (np.abs(g[6] - r[6]) > F32_TOL_STEER) | (np.abs(g[7] - r[7]) > F32_TOL_ACCEL) | (np.abs(g[:6] - r[:6]).max(0) > F32_TOL_STATE)
    if got.get("traj") is not None and ref.get("traj") is not None:
        far = far | (np.abs(got["traj"].astype(np.float64) - ref["traj"]).max(0) > F32_TOL_TRAJ)
    far = far & ok
    same_cost = np.abs(g[8] - r[8]) <= 1e-3 * np.maximum(1.0, np.abs(r[8]))
    return far & same_cost, far & ~same_cost


def closed_loop_report(hist, step_status, ref_hist, ref_status):
    """Closed loops (src/test.cpp:79-111) of two solvers side by side: hist [steps, 9, cars] and the status of every solve
    [steps, cars] of each.  EVERY solve is counted.  A car is "comparable" up to and including its first solve that either
    solver did not converge on (what comes back from a failed solve is fed into the next one) and up to its first solve on
    which both converge to different local minima (|d delta0| > 1e-3 with both SUCCESS: the NLP is non-convex, and late in
    a loop the car is far beyond the stretch of road the polynomial was fitted on)."""
    steps, _, cars = hist.shape
    d = np.abs(hist - ref_hist)
    d_steer, d_acc, d_state = d[:, 6], d[:, 7], d[:, :6].max(1)
    bad = (step_status != 0) | (ref_status != 0)
    after_failure = (np.cumsum(bad, axis=0) - bad) > 0                      # strictly after the first failed solve of the car
    both_ok = ~bad & ~after_failure
    branch = both_ok & (d_steer > 1e-3)
    after_branch = np.cumsum(branch, axis=0) > 0                            # the diverging solve and everything behind it
    comparable = both_ok & ~after_branch
    nan0 = lambda x: np.where(np.isfinite(x), x, 1e300)
    q = lambda x: [float(np.quantile(nan0(x), p)) for p in (0.5, 0.99, 0.999, 1.0)]
    allv = ~after_failure & ~bad                                            # all solves with the same kind of input on both sides
    cm = lambda x: float(x[comparable].max()) if comparable.any() else 0.0
    return {"cars": int(cars), "steps": int(steps), "solves_compared": int(steps * cars),
            "status_differs": int(((step_status != ref_status) & ~after_failure).sum()),
            "status_counts": np.bincount(step_status.ravel().astype(np.int64), minlength=5).tolist(),
            "ref_status_counts": np.bincount(ref_status.ravel().astype(np.int64), minlength=5).tolist(),
            "solves_behind_a_failed_solve": int(after_failure.sum()),
            "cars_on_another_local_minimum": int(after_branch.any(0).sum()), "solves_behind_such_a_fork": int(after_branch.sum()),
            "cars_that_start_a_solve_on_a_yaw_bound": None,
            "quantiles": "p50, p99, p99.9, max over every solve both solvers converged on (forked cars included)",
            "d_steer_rad": q(d_steer[allv]), "d_accel": q(d_acc[allv]), "d_state": q(d_state[allv]),
            "comparable_solves": int(comparable.sum()),
            "comparable_max": {"d_steer_rad": cm(d_steer), "d_accel": cm(d_acc), "d_state": cm(d_state)}}
