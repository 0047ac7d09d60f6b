"""Synthetic batches for the workloads BASELINE.json names.

Pure numpy host code: it produces the *inputs* of the batched solve
(state[6,B], coeffs[5,B], yaw_lo[B], yaw_hi[B]) the way the reference's
telemetry handler and MPC::run() would have produced them one at a time
(src/mpc_main.cpp:126-159, src/control/MPC.cpp:329-356).  Nothing here solves.
The PRNG is numpy's counter-based Philox with a fixed seed, so every rank and
every run sees the same stream.
"""
import os

import numpy as np

from . import _abi

DEFAULT_SEED = 1234


def _rng(seed, stream=0):
    return np.random.Generator(np.random.Philox(key=int(seed), counter=[0, 0, 0, int(stream)]))


def load_waypoints(path):
    """lake_track_waypoints.csv: header 'x,y' then 70 rows."""
    return np.loadtxt(path, delimiter=",", skiprows=1)


# -- pieces of the reference restated in vectorised numpy ---------------------

def _table_lookup(xs, table_x, table_y, maxv):
    """Vehicle::computeSpeedTarget / computeYawChangeSpeedLimit (Vehicle.cpp:34-79), vectorised."""
    y = np.abs(xs)
    tx = np.asarray(table_x, dtype=np.float64)
    ty = np.asarray(table_y, dtype=np.float64)
    idx = np.full(y.shape, len(tx), dtype=np.int64)
    for i in range(len(tx) - 1, -1, -1):
        idx = np.where(y <= tx[i], i, idx)
    idx = np.minimum(idx, len(ty) - 1)
    return np.minimum(ty[idx], maxv)


def polyfit_adaptive(px, py, max_fit_order, max_fit_error):
    """RoadGeometry::fit (RoadGeometry.cpp:18-34) for a batch: px, py are [B, npts] in the vehicle
    frame.  Returns coeffs [5, B] zero padded and the number of coefficients used [B]."""
    B, npts = px.shape
    coeffs = np.zeros((_abi.NCOEF, B))
    ncoef = np.zeros(B, dtype=np.int64)
    fiterr = np.zeros(B)
    pending = np.ones(B, dtype=bool)
    order = 2
    while True:
        used = order
        order += 1
        A = np.stack([px ** j for j in range(used + 1)], axis=2)          # Vandermonde, utils.cpp:15-25
        q, r = np.linalg.qr(A)                                              # Householder QR, utils.cpp:27
        c = np.linalg.solve(r, np.einsum("bij,bi->bj", q, py)[..., None])[..., 0]
        fit = np.einsum("bij,bj->bi", A, c)
        err = np.sum((py - fit) ** 2, axis=1)
        take = pending
        coeffs[:used + 1, take] = c[take].T
        coeffs[used + 1:, take] = 0.0
        ncoef[take] = used + 1
        fiterr[take] = err[take]
        pending = pending & (err > max_fit_error)
        if not (pending.any() and order < max_fit_order):
            break
    return coeffs, ncoef, fiterr


def _polyder_atan(coeffs, x):
    d = np.zeros_like(x)
    for i in range(_abi.NCOEF - 1, 0, -1):
        d = d * x + i * coeffs[i]
    return np.arctan(d)


def run_preprocess(params, pose, ptsx, ptsy):
    """The pre-solve half of MPC::run() (MPC.cpp:329-356) for a batch.
    pose: dict of arrays x,y,psi,v,steer [B]; ptsx/ptsy: [B, npts] global frame."""
    cos, sin = np.cos(pose["psi"])[:, None], np.sin(pose["psi"])[:, None]
    vx = ptsx - pose["x"][:, None]
    vy = ptsy - pose["y"][:, None]
    lx = vx * cos + vy * sin                                                # Vehicle.cpp:105-114
    ly = vy * cos - vx * sin
    coeffs, ncoef, fiterr = polyfit_adaptive(lx, ly, params.max_fit_order, params.max_fit_error)
    cte = coeffs[0].copy()                                                  # f(0), MPC.cpp:334
    epsi = -np.arctan(coeffs[1])                                            # MPC.cpp:336
    back, front = lx[:, -1], lx[:, 0]
    # computeOrientationChange(0, back): dir = back - 0 (RoadGeometry.cpp:57-61); dir < 0 adds pi to both
    # angles before normalising, so the difference is unchanged unless the wrap differs; restated exactly:
    def orient(px_, dir_):
        psi = _polyder_atan(coeffs, px_)
        adj = psi + np.pi
        adj = np.where(adj >= np.pi, adj - 2 * np.pi, adj)
        return np.where(dir_ < 0, adj, psi)
    max_yaw_change = (orient(back, back) - orient(np.zeros_like(back), back)) * (back - front) / back
    yc = [params.yaw_changes[i] for i in range(params.n_yaw_changes)]
    ycs = [params.yaw_change_speeds[i] for i in range(params.n_yaw_change_speeds)]
    max_speed = _table_lookup(max_yaw_change, yc, ycs, params.max_speed)    # MPC.cpp:340
    st = [params.steers[i] for i in range(params.n_steers)]
    sts = [params.steer_speeds[i] for i in range(params.n_steer_speeds)]
    target_speed = np.minimum(_table_lookup(pose["steer"], st, sts, np.inf), max_speed)  # MPC.cpp:342
    yaw_lo = np.where(max_yaw_change < 0, max_yaw_change, -0.1)             # MPC.cpp:345-352
    yaw_hi = np.where(max_yaw_change < 0, 0.1, max_yaw_change)
    B = len(cte)
    state = np.zeros((_abi.NSTATE, B))
    state[3] = pose["v"]; state[4] = cte; state[5] = epsi                   # MPC.cpp:355-356
    return {"state": state, "coeffs": coeffs, "ncoef": ncoef, "fiterr": fiterr, "yaw_lo": yaw_lo, "yaw_hi": yaw_hi,
            "max_yaw_change": max_yaw_change, "max_speed": max_speed, "target_speed": target_speed,
            "ptsx_vehicle": lx, "ptsy_vehicle": ly}


# -- BASELINE.json configs ----------------------------------------------------

def straight_line_batch(B, params, seed=DEFAULT_SEED):
    """configs[1]: straight reference line with a lateral offset c0 ~ U(-2,2) m and a heading error
    theta ~ U(-0.2,0.2) rad (c1 = tan(theta)), v ~ U(5,45) m/s, yaw bounds [-0.1, +0.1]."""
    g = _rng(seed, 2)
    c0 = g.uniform(-2.0, 2.0, B)
    theta = g.uniform(-0.2, 0.2, B)
    v = np.minimum(g.uniform(5.0, 45.0, B), 0.98 * params.max_speed)
    coeffs = np.zeros((_abi.NCOEF, B))
    coeffs[0] = c0
    coeffs[1] = np.tan(theta)
    state = np.zeros((_abi.NSTATE, B))
    state[3] = v; state[4] = c0; state[5] = -np.arctan(coeffs[1])
    return {"state": state, "coeffs": coeffs, "yaw_lo": np.full(B, -0.1), "yaw_hi": np.full(B, 0.1)}


def lake_track_batch(B, params, waypoints, seed=DEFAULT_SEED, stream=3, latency_s=None, npts=6, filtered=True):
    """configs[2]: car placed on a random segment of the lake track, latency-compensated exactly as
    mpc_main.cpp:155-159 (mean solver time fixed to 0), next 6 waypoints, then MPC::run() preprocessing.
    Instances are redrawn when the reference's own road model does not hold for them: compensated speed
    above max_speed (the reference's NLP is infeasible), psi = 0 on a yaw bound, waypoints that are not
    strictly increasing in the vehicle-frame x (y = f(x) is then not a function: the track doubles back
    inside the 6-point window), or a final fit error above Config::maxFitError (RoadGeometry.cpp:34).
    The returned dict carries the counts ("drawn", "rejected": per criterion) so that callers can report them;
    filtered=False keeps every draw with a finite fit (the unfiltered population: the solver then reports the
    infeasible / non-converged ones through its per-instance status); filtered="survey" applies exactly the rejection
    SURVEY.md section 8d names for this generator -- compensated speed above Config::maxSpeed (the reference's NLP has no
    feasible point then) -- and nothing else: waypoint windows that double back and fits above Config::maxFitError stay
    in, as they would reach the reference's solver (RoadGeometry.cpp:26-34 stops raising the order and solves anyway)."""
    wp = np.asarray(waypoints, dtype=np.float64)
    nwp = len(wp)
    g = _rng(seed, stream)
    lat = params.lookahead if latency_s is None else latency_s
    keys = ("state", "coeffs", "yaw_lo", "yaw_hi", "ncoef", "max_yaw_change", "target_speed", "v0", "pose", "ptsx", "ptsy")
    acc = {k: [] for k in keys}
    need = B
    drawn = 0
    rej = {"speed_above_max": 0, "psi0_on_yaw_bound": 0, "waypoints_not_monotone_in_x": 0, "fit_error_above_max": 0, "nonfinite_fit": 0}
    while need > 0:
        n = max(int(need * 1.3) + 16, 64)
        k = g.integers(0, nwp, n)
        frac = g.uniform(0.0, 1.0, n)
        lateral = g.uniform(-1.5, 1.5, n)
        dpsi = g.uniform(-0.15, 0.15, n)
        v = g.uniform(10.0, 60.0, n)
        steer = g.uniform(-0.2, 0.2, n)
        throttle = g.uniform(-0.2, 1.0, n)
        a, b = wp[k], wp[(k + 1) % nwp]
        seg = b - a
        heading = np.arctan2(seg[:, 1], seg[:, 0])
        nx, ny = -np.sin(heading), np.cos(heading)
        px = a[:, 0] + frac * seg[:, 0] + lateral * nx
        py = a[:, 1] + frac * seg[:, 1] + lateral * ny
        psi = heading + dpsi
        psi = np.where(psi >= np.pi, psi - 2 * np.pi, psi)                  # normalizeAngle, mpc_main.cpp:127
        psi = np.where(psi < -np.pi, psi + 2 * np.pi, psi)
        accel = (throttle - v / 50.0) * 6                                   # mpc_main.cpp:156
        if lat > 0:                                                         # Vehicle::move, Vehicle.cpp:145-168
            dist = v * lat
            npsi = psi + steer * dist / params.Lf
            px = px + dist * np.cos(psi)
            py = py + dist * np.sin(psi)
            v = v + accel * lat
            psi = npsi
        idx = (k[:, None] + np.arange(npts)[None, :]) % nwp
        pre = run_preprocess(params, {"x": px, "y": py, "psi": psi, "v": v, "steer": steer}, wp[idx, 0], wp[idx, 1])
        c_speed = np.abs(v) < params.max_speed
        c_yaw = (pre["yaw_lo"] < -1e-3) & (pre["yaw_hi"] > 1e-3)
        c_fin = np.isfinite(pre["coeffs"]).all(axis=0)
        c_mono = (np.diff(pre["ptsx_vehicle"], axis=1) > 0).all(axis=1)
        c_fit = pre["fiterr"] <= params.max_fit_error
        ok = (c_speed & c_fin) if filtered == "survey" else ((c_speed & c_yaw & c_fin & c_mono & c_fit) if filtered else c_fin)
        take = np.flatnonzero(ok)[:need]
        used = (take[-1] + 1) if len(take) == need and len(take) else n      # draws looked at in this round
        drawn += int(used)
        for name, c in (("speed_above_max", c_speed), ("psi0_on_yaw_bound", c_yaw), ("waypoints_not_monotone_in_x", c_mono),
                        ("fit_error_above_max", c_fit), ("nonfinite_fit", c_fin)):
            rej[name] += int((~c[:used]).sum())
        acc["state"].append(pre["state"][:, take]); acc["coeffs"].append(pre["coeffs"][:, take])
        for kk in ("yaw_lo", "yaw_hi", "ncoef", "max_yaw_change", "target_speed"):
            acc[kk].append(pre[kk][take])
        acc["v0"].append(v[take])
        # the raw telemetry MPC::run() receives: latency-compensated pose {x,y,psi,v,steering,acceleration}
        # and the 6 global waypoints (struct-of-arrays, instance last)
        acc["pose"].append(np.stack([px, py, psi, v, steer, accel])[:, take])
        acc["ptsx"].append(wp[idx, 0].T[:, take]); acc["ptsy"].append(wp[idx, 1].T[:, take])
        need -= len(take)
    two_d = ("state", "coeffs", "pose", "ptsx", "ptsy")
    out = {k: np.concatenate(acc[k], axis=1 if k in two_d else 0) for k in keys}
    for k in two_d:
        out[k] = np.ascontiguousarray(out[k])
    out["drawn"] = drawn
    out["rejected"] = ({"speed_above_max": rej["speed_above_max"], "nonfinite_fit": rej["nonfinite_fit"]} if filtered == "survey" else
                       rej if filtered else {"nonfinite_fit": rej["nonfinite_fit"]})
    return out


def weight_sweep(B, params, seed=DEFAULT_SEED, velocity_weights=(1.0, 100.0)):
    """configs[4]: per-instance Config::weights with the values swept in submission-report.md:303-319.
    The report also sweeps the velocity weight to 0 ("the vehicle decelerates", examples/velocity-weights.png): pass
    velocity_weights=(0.0, 1.0, 100.0) for that.  It is left out of the default sweep because with no velocity term
    the acceleration is bang-bang on a cost difference of ~1e-6: fp64 reproduces the reference's choice (tests), the
    fp32 mode cannot be held to an a0 tolerance there."""
    g = _rng(seed, 5)
    w = np.tile(np.array([params.weights[i] for i in range(_abi.NW)])[:, None], (1, B))
    w[1] = g.choice([80.0, 100.0, 120.0, 1000.0], B)
    w[2] = g.choice(list(velocity_weights), B)
    w[3] = g.choice([1.0, 300.0, 5000.0], B)
    w[6] = g.choice([0.0, 100.0, 10000.0], B)
    return np.ascontiguousarray(w)


def golden_dir():
    return os.path.join(_abi.ROOT, "tests", "golden")
