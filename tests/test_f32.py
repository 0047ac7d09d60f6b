"""MPC_PRECISION_F32 (BASELINE.json configs[4]: "fp32 mixed precision, weight sweep") against the fp64 oracle, every instance's
status accounted for.  Two modes (MpcParams.f32_finish):
  1 (default)  fp32 interior-point iterations down to the barrier parameter mixed_switch_mu, then every instance finished in
               fp64 (same state machine, tol = 1e-8, polish); fp32 at the ABI.  Tolerances helpers.F32_TOL_* (1e-3 on delta0,
               a0 and the step-1 state, 1e-2 m on the trajectory), stated before measuring; velocity weight 0 and long horizons
               included; the status of an instance is the fp64 solver's.
  0            the pure fp32 solver of round 2, tolerances helpers.F32PURE_TOL_*.
The CPU tests replay both with the test-only host build of the same header (tests/host_twin); the `gpu` tests run the HIP
kernels through mpc_solve_batch_device_f32 / mpc_solve_batch_host_f32."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from helpers import (F32_TOL_ACCEL, F32_TOL_COST_REL, F32_TOL_STATE, F32_TOL_STEER, F32_TOL_TRAJ, F32PURE_TOL_ACCEL, F32PURE_TOL_COST_REL,
                     F32PURE_TOL_STATE, F32PURE_TOL_STEER, F32PURE_TOL_TRAJ, f32_forks, oracle_solve_batch, twin_solve, twin_solve_f32, twin_solve_mixed,
                     twin_solve_mixed_f64, vp)

ZERO_V = (0.0, 1.0, 100.0)      # the velocity weights of SURVEY.md 8d Config 5 (submission-report.md:315: "the vehicle decelerates")


def _f32_params(pkg, golden_dir, config="config-fast.json", finish=1, **over):
    p = pkg.params_from_json(os.path.join(golden_dir, config), **over)
    p.precision = pkg.PRECISION_F32
    p.f32_finish = finish
    return p


def _check_f32(got, ref, what, n_expected, pure=False):
    """got: fp32 results (every instance), ref: fp64 results of the same instances.  100 % status accounting: in the default
    mode an instance ends with the status the fp64 solver gives it; the pure fp32 solver must converge wherever fp64 does."""
    assert len(got["status"]) == n_expected
    if pure:
        assert (ref["status"] == 0).all(), (what, np.bincount(ref["status"]))
        assert (got["status"] == 0).all(), (what, "fp32 statuses", np.bincount(got["status"]), np.where(got["status"] != 0)[0][:8])
    else:
        assert np.array_equal(got["status"], ref["status"]), (what, np.bincount(got["status"]), np.bincount(ref["status"]), np.where(got["status"] != ref["status"])[0][:8])
    ok = (got["status"] == 0) & (ref["status"] == 0)
    if not pure:
        # other local minima (helpers.f32_forks): counted, bounded at one in 10 000, excluded from the comparison of the numbers
        fork, wrong = f32_forks(got, ref, ok)
        assert fork.sum() <= max(2, n_expected // 10000) and wrong.sum() == 0, (what, "other local minima:", np.where(fork)[0][:8], "wrong:", np.where(wrong)[0][:8])
        ok = ok & ~fork
    ts, ta, tx, tt, tc = ((F32PURE_TOL_STEER, F32PURE_TOL_ACCEL, F32PURE_TOL_STATE, F32PURE_TOL_TRAJ, F32PURE_TOL_COST_REL) if pure else
                          (F32_TOL_STEER, F32_TOL_ACCEL, F32_TOL_STATE, F32_TOL_TRAJ, F32_TOL_COST_REL))
    g = got["out"].astype(np.float64)
    d = np.abs(g - ref["out"])[:, ok]
    rel_cost = d[8] / np.maximum(1.0, np.abs(ref["out"][8][ok]))
    assert d[6].max() <= ts, "%s max |d steer| = %g" % (what, d[6].max())
    assert d[7].max() <= ta, "%s max |d accel| = %g" % (what, d[7].max())
    assert d[:6].max() <= tx, "%s max |d step-1 state| = %g" % (what, d[:6].max())
    assert rel_cost.max() <= tc, "%s max rel |d cost| = %g" % (what, rel_cost.max())
    if pure:
        assert np.quantile(d[6], 0.99) <= 5e-4 and np.quantile(d[6], 0.999) <= 2e-3 and np.quantile(d[7], 0.99) <= 2e-3, what
    if got.get("traj") is not None and ref.get("traj") is not None:
        dt_ = np.abs(got["traj"].astype(np.float64) - ref["traj"])[:, ok]
        assert dt_.max() <= tt, (what, dt_.max())
    return d


def test_f32_finish_twin_weight_sweep_incl_zero_velocity_weight(pkg, host_twin, golden_dir, waypoints):
    """configs[4] at CPU test size, the shipped mode: lake-track states + per-instance weights -- all the values swept in the
    reference's submission-report.md:303-319, velocity weight 0 included -- against the dense fp64 oracle."""
    params = _f32_params(pkg, golden_dir)
    B = 192
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=51)
    w = pkg.scenarios.weight_sweep(B, params, seed=52, velocity_weights=ZERO_V)
    assert (w[2] == 0).sum() > 40
    r = twin_solve_mixed(host_twin, params, b, weights=w)
    ref = oracle_solve_batch(O.load_config("config-fast.json"), b, range(B), weights=w)
    _check_f32(r, ref, "mixed weight sweep", B)
    assert (r["iters_f32"] > 0).all() and (r["iters"] > r["iters_f32"]).all()        # both phases did their part
    w2 = w.copy(); w2[6] = 4321.0                  # the acceleration weight has no effect under the frozen tape
    r2 = twin_solve_mixed(host_twin, params, b, weights=w2)
    assert np.array_equal(r2["out"], r["out"])


@pytest.mark.parametrize("config,N,dt,B", [("config-fast.json", 10, 0.1, 4096), ("config-stable.json", 25, 0.05, 1024), ("config-fast.json", 40, 0.025, 512)])
def test_f32_finish_twin_matches_fp64_twin_at_scale(pkg, host_twin, golden_dir, waypoints, config, N, dt, B):
    """The shipped mode against the fp64 build of the same solver (pinned to the oracle at 1e-6 by the other tests): weight sweeps
    with velocity weight 0, N = 10 / 25 / 40 -- long horizons are no longer "best effort": same status as fp64 on every instance,
    the stated tolerances on every instance both converge on, and about as many iterations in total as the fp64 solve."""
    params = _f32_params(pkg, golden_dir, config=config, N=N, dt=dt)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=61)
    w = pkg.scenarios.weight_sweep(B, params, seed=44, velocity_weights=ZERO_V)
    r64 = twin_solve(host_twin, p64, b, weights=w)
    r = twin_solve_mixed(host_twin, params, b, weights=w)
    _check_f32(r, r64, "mixed vs twin64 (N=%d)" % N, B)
    # (an instance that uses up the fp32 phase's 16 iterations, or leaves it out of trouble, starts again in fp64: 12 % of a weight sweep)
    assert abs(r["iters"].mean() - r64["iters"].mean()) < 2.0
    assert r["iters_f32"].mean() > 0.45 * r["iters"].mean()                          # about half of the iterations run in fp32


def test_f64_f32_start_auto_for_long_horizons_and_the_single_phase_verdict(pkg, host_twin, golden_dir, waypoints):
    """MpcParams.f64_f32_start = MPC_F32_START_AUTO (the default): fp64 handles with N >= 15 run their early iterations on the fp32
    record.  The whole N = 25 batch of the full-size soak (8 192 instances) through the host replay of the two phases: same status
    as the single-phase solve on every instance, outputs within the fp64 tolerances -- including the instances the fp64 phase
    cannot finish from where fp32 left them (a failed line search on the device before this rule existed): those are solved again
    from the start point exactly as the single-phase solve does it, and so return its status and its point."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"), N=25, dt=0.05)
    assert params.f64_f32_start == 2                      # MPC_F32_START_AUTO, as shipped: on from N = 15
    B = 8192
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=103)
    r = twin_solve_mixed_f64(host_twin, params, b)
    r0 = twin_solve(host_twin, params, b)
    assert np.array_equal(r["status"], r0["status"]) and (r["status"] == 0).all()
    d = np.abs(r["out"][:8] - r0["out"][:8])
    assert d[6].max() <= 1e-6 and d[7].max() <= 1e-6 and d[:6].max() <= 1e-5 and np.abs(r["traj"] - r0["traj"]).max() <= 1e-5
    assert (r["iters_f32"] > 0).all() and (r["iters_f32"] <= 16).all()
    assert abs(r["iters"].mean() - r0["iters"].mean()) < 0.5 and r["iters"].max() <= r0["iters"].max() + 60


def test_hard_instances_of_the_unfiltered_population_single_phase_and_f32_start(pkg, host_twin, golden_dir, waypoints):
    """configs[3]'s share drawn with SURVEY 8d's rejection only (32 768 instances, N = 25) has hard instances of 50-170 iterations on
    which fp32 iterates lead elsewhere than fp64 ones: continued by the fp64 solver from wherever the fp32 phase stopped, five of
    the seven below ended in another local minimum and two converged where the oracle reports a failed line search.  Hence the
    rule that only a CLEAN hand-over is continued (barrier parameter at its switch value, or tol_f32 met) and everything else --
    allowance used up, line search or inertia correction out of single precision -- is solved in fp64 from the start point.
    Asserted: the single-phase solve (the default) and the fp32 start both give the oracle's status, iteration count (fp64 part)
    and point on all seven."""
    params = pkg.params_from_json(os.path.join(golden_dir, "config-stable.json"), N=25, dt=0.05)
    b = pkg.scenarios.lake_track_batch(32768, params, waypoints, stream=3, filtered="survey")
    hard = [6974, 7225, 8421, 15960, 18245, 18943, 25724]
    sub = {k: np.ascontiguousarray(b[k][..., hard]) for k in ("state", "coeffs", "yaw_lo", "yaw_hi")}
    r0 = twin_solve(host_twin, params, sub)
    ref = oracle_solve_batch(O.load_config("config-stable.json", N=25, dt=0.05), sub, range(len(hard)), opt=O.default_options(max_iter=params.max_iter))
    assert list(ref["status"]) == [2, 0, 0, 0, 2, 0, 0]
    assert np.array_equal(r0["status"], ref["status"]) and np.abs(r0["iters"] - ref["iters"]).max() <= 1
    assert np.abs(r0["out"][:8] - ref["out"][:8]).max() <= 1e-6
    q = params.copy(); q.f64_f32_start = 2
    r = twin_solve_mixed_f64(host_twin, q, sub)
    assert np.array_equal(r["status"], ref["status"])
    assert np.abs(r["out"][:8] - ref["out"][:8]).max() <= 1e-6
    assert np.abs((r["iters"] - r["iters_f32"]) - ref["iters"]).max() <= 1 and (r["iters_f32"] <= 16).all()      # solved from the start point in fp64


def test_f32_pure_twin_weight_sweep_matches_fp64_oracle(pkg, host_twin, golden_dir, waypoints):
    """f32_finish = 0: the pure fp32 solver against the dense fp64 oracle (its own tolerances; velocity weights 1 and 100)."""
    params = _f32_params(pkg, golden_dir, finish=0)
    B = 192
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=51)
    w = pkg.scenarios.weight_sweep(B, params, seed=52)
    r = twin_solve_f32(host_twin, params, b, weights=w)
    ref = oracle_solve_batch(O.load_config("config-fast.json"), b, range(B), weights=w)
    _check_f32(r, ref, "weight sweep", B, pure=True)
    # the acceleration weight has no effect under the frozen tape: bitwise, also in fp32
    w2 = w.copy(); w2[6] = 4321.0
    r2 = twin_solve_f32(host_twin, params, b, weights=w2)
    assert np.array_equal(r2["out"], r["out"])


def test_f32_pure_twin_matches_fp64_twin_at_scale(pkg, host_twin, golden_dir, waypoints):
    """f32_finish = 0: 4096 + 4096 instances (plain and weight sweep) against the fp64 build of the same solver: every instance
    converges in fp32 and stays inside the pure mode's tolerances."""
    params = _f32_params(pkg, golden_dir, finish=0)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    for seed, weights in ((41, False), (43, True)):
        b = pkg.scenarios.lake_track_batch(4096, params, waypoints, seed=seed)
        w = pkg.scenarios.weight_sweep(4096, params, seed=44) if weights else None
        r64 = twin_solve(host_twin, p64, b, weights=w, want_traj=False)
        r32 = twin_solve_f32(host_twin, params, b, weights=w, want_traj=False)
        _check_f32(r32, r64, "twin32 vs twin64 (weights=%s)" % weights, 4096, pure=True)
        assert r32["iters"].mean() < r64["iters"].mean()        # the looser tolerance is also fewer iterations


@pytest.mark.parametrize("config,N,dt", [("config-stable.json", 25, 0.05), ("config-fast.json", 40, 0.025)])
def test_f32_pure_long_horizons_are_best_effort(pkg, host_twin, golden_dir, waypoints, config, N, dt):
    """f32_finish = 0.  The pure fp32 solver is specified for BASELINE.json configs[4] (N = 10).  On long horizons (x reaches 60 m, 150-240
    constraints) it still works, with this accounting: a few instances per thousand end at the noise floor of single
    precision without meeting the fp32 tolerance -- the constraint violation cannot go below ~1e-5 there and a Newton step
    computed from that noise can throw the duals out -- and are REPORTED as such (status LINESEARCH or MAXITER, never
    silently wrong); everything reported as solved is within looser tolerances of the fp64 build: steer 2e-3 rad, step-1
    state 2e-2.  (A guard that refuses such steps was tried: it rescues those instances and costs every other one a factor
    of two in accuracy, so it is not in.)"""
    params = _f32_params(pkg, golden_dir, config=config, finish=0, N=N, dt=dt)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    B = 1024
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=61)
    r64 = twin_solve(host_twin, p64, b, want_traj=False)
    r32 = twin_solve_f32(host_twin, params, b, want_traj=False)
    assert set(np.unique(r32["status"])) <= {0, 1, 2}
    assert (r32["status"] != 0).sum() <= 6, np.bincount(r32["status"])
    ok = (r32["status"] == 0) & (r64["status"] == 0)
    d = np.abs(r32["out"].astype(np.float64) - r64["out"])[:, ok]
    assert d[6].max() <= 2e-3 and d[:6].max() <= 2e-2, (d[6].max(), d[:6].max())
    assert np.quantile(d[6], 0.99) <= 5e-4


def test_f32_math_kernels(host_twin):
    """sin/cos/atan of the fp32 solver (polynomial kernels in mpc_core.h)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-4, 4, 20000), rng.uniform(-60, 60, 20000), rng.normal(0, 1e-3, 2000)]).astype(np.float32)
    sn = np.zeros_like(x); cs = np.zeros_like(x); at = np.zeros_like(x)
    host_twin.mpc_host_twin_math_f32(C.c_int64(len(x)), vp(x), vp(sn), vp(cs), vp(at))
    xd = x.astype(np.float64)
    assert np.max(np.abs(sn - np.sin(xd))) < 2.5e-7 and np.max(np.abs(cs - np.cos(xd))) < 2.5e-7
    assert np.max(np.abs(at - np.arctan(xd))) < 2.5e-7


def test_f32_edge_cases_twin(pkg, host_twin, golden_dir):
    """Infeasible fixed initial state is flagged in fp32 too; the straight road with zero errors gives zero steering."""
    params = _f32_params(pkg, golden_dir, "config-stable.json")
    st = np.zeros((6, 3)); cf = np.zeros((5, 3))
    st[3] = [60.0, 20.0, 20.0]; st[2, 1] = 0.2
    b = {"state": st, "coeffs": cf, "yaw_lo": np.full(3, -0.1), "yaw_hi": np.full(3, 0.1)}
    r = twin_solve_f32(host_twin, params, b)
    assert list(r["status"]) == [3, 3, 0] and abs(r["out"][6, 2]) < 1e-5


def _gpu_solve(pkg, params, b, w, dt, want_traj=True):
    import torch
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    B = b["state"].shape[1]
    with pkg.BatchedMPC(params, B, device=0) as mpc:
        r = mpc.solve_torch(t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]), weights=t(w) if w is not None else None, want_traj=want_traj)
        torch.cuda.synchronize()
        st = mpc.stats()
        res = {k: (v.cpu().numpy() if v is not None else None) for k, v in r.items()}
        assert st.batch == B and st.n_success == int((res["status"] == 0).sum())
    return res


@pytest.mark.gpu
def test_f32_gpu_weight_sweep_matches_oracle(pkg, host_twin, golden_dir, waypoints):
    """BASELINE.json configs[4] on the device, the shipped mode (fp32 phase + fp64 finish) through mpc_solve_batch_device_f32:
    16 384 instances with per-instance weights, velocity weight 0 included; ALL of them against the fp64 HIP path (itself pinned
    to the oracle at 1e-6 in test_gpu_parity.py), a sample against the dense fp64 oracle, and the host replay of the two phases."""
    import torch
    params = _f32_params(pkg, golden_dir)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    B = 16384
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=53)
    w = pkg.scenarios.weight_sweep(B, params, seed=54, velocity_weights=ZERO_V)
    r32 = _gpu_solve(pkg, params, b, w, torch.float32)
    r64 = _gpu_solve(pkg, p64, b, w, torch.float64)
    assert r32["out"].dtype == np.float32 and r32["traj"].dtype == np.float32                    # fp32 at the ABI
    _check_f32(r32, r64, "HIP mixed vs HIP fp64", B)
    idx = list(range(0, B, B // 96))
    ref = oracle_solve_batch(O.load_config("config-fast.json"), b, idx, weights=w)
    _check_f32({k: (v[..., idx] if v is not None else None) for k, v in r32.items()}, ref, "HIP mixed vs oracle", len(idx))
    # the host replay of the same two phases takes the same path up to the rounding of the device's own math
    rt = twin_solve_mixed(host_twin, params, {k: b[k][..., :2048] for k in ("state", "coeffs", "yaw_lo", "yaw_hi")}, weights=w[:, :2048], want_traj=False)
    assert np.array_equal(rt["status"], r32["status"][:2048])
    assert np.abs(rt["out"][:8].astype(np.float64) - r32["out"][:8, :2048]).max() <= 1e-4
    assert abs(rt["iters"].mean() - r32["iters"][:2048].mean()) < 0.2
    # an fp32 handle refuses the fp64 entry point (and says why)
    with pkg.BatchedMPC(params, 16, device=0) as mpc:
        z = torch.zeros(16, dtype=torch.float64, device="cuda:0")
        rc = pkg.library().mpc_solve_batch_device(mpc._h, 1, 1, *([C.c_void_p(z.data_ptr())] * 9), None)
        assert rc == -1 and b"precision" in pkg.library().mpc_last_error()
        # ... and the host entry point of its own precision gives the device path's answers bit for bit
        sub = {k: b[k][..., :777] for k in ("state", "coeffs", "yaw_lo", "yaw_hi")}
        rh = mpc.solve_numpy(sub["state"], sub["coeffs"], sub["yaw_lo"], sub["yaw_hi"], weights=w[:, :777], want_traj=True) if False else None
    with pkg.BatchedMPC(params, 777, device=0) as mpc:
        rh = mpc.solve_numpy(b["state"][:, :777], b["coeffs"][:, :777], b["yaw_lo"][:777], b["yaw_hi"][:777], weights=w[:, :777], want_traj=True)
    for k in ("out", "traj", "status", "iters"):
        assert np.array_equal(rh[k], r32[k][..., :777]), k


@pytest.mark.gpu
def test_f32_pure_gpu_weight_sweep_matches_oracle(pkg, host_twin, golden_dir, waypoints):
    """f32_finish = 0 on the device: the pure fp32 solver, both builds of the kernel (one and two waves per SIMD) bitwise equal,
    all instances against the fp64 HIP path at the pure mode's tolerances."""
    import torch
    params = _f32_params(pkg, golden_dir, finish=0)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    B = 16384
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=53)
    w = pkg.scenarios.weight_sweep(B, params, seed=54, velocity_weights=(1.0, 100.0))
    res = {}
    old = os.environ.get("MPC_F32_OCC")
    try:
        for occ in ("2", "1"):
            os.environ["MPC_F32_OCC"] = occ
            res[occ] = _gpu_solve(pkg, params, b, w, torch.float32)
    finally:
        if old is None:
            os.environ.pop("MPC_F32_OCC", None)
        else:
            os.environ["MPC_F32_OCC"] = old
    for k in ("out", "traj", "status", "iters"):
        assert np.array_equal(res["2"][k], res["1"][k]), k
    r32 = res["2"]
    r64 = _gpu_solve(pkg, p64, b, w, torch.float64)
    _check_f32(r32, r64, "HIP fp32 vs HIP fp64", B, pure=True)
    rt = twin_solve_f32(host_twin, params, {k: b[k][..., :2048] for k in ("state", "coeffs", "yaw_lo", "yaw_hi")}, weights=w[:, :2048], want_traj=False)
    assert (rt["status"] == 0).all()
    assert np.abs(rt["out"][6].astype(np.float64) - r32["out"][6, :2048]).max() <= F32PURE_TOL_STEER


@pytest.mark.gpu
@pytest.mark.parametrize("finish", [1, 0])
def test_f32_gpu_plain_batch_and_edges(pkg, golden_dir, waypoints, finish):
    """fp32 without per-instance weights, ragged batch size, status codes, empty batch; both modes."""
    import torch
    dev = torch.device("cuda:0")
    params = _f32_params(pkg, golden_dir, finish=finish)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    B = 4096 + 37
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=55)
    b["state"][3, 5] = 99.0                                                   # infeasible: v0 above max_speed
    b["state"][5, 9] = np.nan                                                 # epsi0 not a number: reported by the phase that meets it
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    outs = {}
    for p, dt in ((params, torch.float32), (p64, torch.float64)):
        with pkg.BatchedMPC(p, B, device=0) as mpc:
            r = mpc.solve_torch(t(b["state"], dt), t(b["coeffs"], dt), t(b["yaw_lo"], dt), t(b["yaw_hi"], dt), want_traj=True)
            torch.cuda.synchronize()
            outs[dt] = {k: v.cpu().numpy() for k, v in r.items()}
            e = mpc.solve_torch(t(np.zeros((6, 0)), dt), t(np.zeros((5, 0)), dt), t(np.zeros(0), dt), t(np.zeros(0), dt))
            assert e["out"].shape == (9, 0)
    r32, r64 = outs[torch.float32], outs[torch.float64]
    assert r32["status"][5] == 3 and r64["status"][5] == 3 and r32["status"][9] == 4 and r64["status"][9] == 4
    keep = (np.arange(B) != 5) & (np.arange(B) != 9)
    _check_f32({k: v[..., keep] for k, v in r32.items()}, {k: v[..., keep] for k, v in r64.items()}, "plain batch", B - 2, pure=not finish)


@pytest.mark.gpu
def test_f32_full_size_properties_config4_shard(pkg, golden_dir, waypoints):
    """BASELINE.json configs[4] at the size ONE GPU gets (1 048 576 / 8 = 131 072 instances, fp32 mixed precision, per-instance
    weights with velocity weight 0, no trajectories): EVERY instance against the fp64 HIP path at the stated tolerances with the
    same status, size-independent properties, and a sample against the fp64 oracle."""
    import torch
    dev = torch.device("cuda:0")
    params = _f32_params(pkg, golden_dir)
    p64 = params.copy(); p64.precision = pkg.PRECISION_F64
    B = 131072
    b = pkg.scenarios.lake_track_batch(B, params, waypoints, seed=56)
    w = pkg.scenarios.weight_sweep(B, params, seed=57, velocity_weights=ZERO_V)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=torch.float32)
    with pkg.BatchedMPC(params, B, device=0) as mpc:
        ins = [t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"])]
        r = mpc.solve_torch(*ins, weights=t(w)); torch.cuda.synchronize()
        st = mpc.stats()
        out = r["out"].cpu().numpy().astype(np.float64); status = r["status"].cpu().numpy(); iters = r["iters"].cpu().numpy()
        r2 = mpc.solve_torch(*ins, weights=t(w)); torch.cuda.synchronize()
        assert torch.equal(r2["out"], r["out"]) and torch.equal(r2["status"], r["status"])            # deterministic
        perm = np.random.default_rng(3).permutation(B)
        rp = mpc.solve_torch(t(b["state"][:, perm]), t(b["coeffs"][:, perm]), t(b["yaw_lo"][perm]), t(b["yaw_hi"][perm]), weights=t(w[:, perm]))
        torch.cuda.synchronize()
        assert np.array_equal(rp["out"].cpu().numpy(), r["out"].cpu().numpy()[:, perm])                # instances are independent
    counts = np.bincount(status, minlength=5)
    assert counts.sum() == B and (st.n_success, st.n_maxiter, st.n_linesearch) == (int(counts[0]), int(counts[1]), int(counts[2]))
    assert counts[3] == 0 and counts[4] == 0 and counts[0] >= B - 16, counts
    r64 = _gpu_solve(pkg, p64, b, w, torch.float64, want_traj=False)
    _check_f32({"out": r["out"].cpu().numpy(), "status": status, "traj": None}, r64, "configs[4] share, every instance", B)
    ok = status == 0
    s0, cf = b["state"].astype(np.float32).astype(np.float64), b["coeffs"].astype(np.float32).astype(np.float64)
    dt, Lf = params.dt, params.Lf
    assert np.all(np.abs(out[6]) <= params.max_steering + 1e-6) and np.all(out[7, ok] <= params.max_acceleration + 1e-5)
    assert np.all(out[7, ok] >= params.max_deceleration - 1e-5)
    # step-1 state satisfies the model equations (MPC.cpp:142-152) from the fixed initial state, to the rounding of the fp32 outputs
    v0 = s0[3]
    assert np.max(np.abs(out[0] - v0 * dt)[ok]) < 1e-5 and np.max(np.abs(out[1])[ok]) < 1e-5
    assert np.max(np.abs(out[2] - out[6] * v0 * dt / Lf)[ok]) < 1e-5
    assert np.max(np.abs(out[3] - (v0 + out[7] * dt))[ok]) < 1e-4
    assert np.max(np.abs(out[4] - (cf[0] + np.sin(s0[5]) * v0 * dt))[ok]) < 1e-4
    idx = [int(i) for i in np.random.default_rng(4).choice(np.where(ok)[0], 128, replace=False)]
    ref = oracle_solve_batch(O.load_config("config-fast.json"), b, idx, weights=w)
    assert (ref["status"] == 0).all()
    d = np.abs(out[:, idx] - ref["out"])
    assert d[6].max() <= F32_TOL_STEER and d[7].max() <= F32_TOL_ACCEL and d[:6].max() <= F32_TOL_STATE
