"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
this module; the product path never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

MAX_TABLE = 16
MAX_COEF = 8


class OrcConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("dt", C.c_double), ("ipopt_timeout", C.c_double), ("latency", C.c_long),
        ("lookahead", C.c_double), ("max_fit_order", C.c_int), ("max_fit_error", C.c_double),
        ("max_steering", C.c_double), ("max_acceleration", C.c_double), ("max_deceleration", C.c_double),
        ("max_speed", C.c_double), ("yaw_low", C.c_double), ("yaw_high", C.c_double),
        ("steer_adj_thresh", C.c_double), ("steer_adj_ratio", C.c_double), ("Lf", C.c_double),
        ("cte_panic", C.c_double), ("epsi_panic", C.c_double), ("n_weights", C.c_int),
        ("weights", C.c_double * MAX_TABLE),
        ("n_steers", C.c_int), ("n_steer_speeds", C.c_int), ("n_yaw_changes", C.c_int),
        ("n_yaw_change_speeds", C.c_int),
        ("steers", C.c_double * MAX_TABLE), ("steer_speeds", C.c_double * MAX_TABLE),
        ("yaw_changes", C.c_double * MAX_TABLE), ("yaw_change_speeds", C.c_double * MAX_TABLE),
    ]


class OrcSolveOptions(C.Structure):
    _fields_ = [("branch_mode", C.c_int), ("max_iter", C.c_int), ("tol", C.c_double),
                ("lam_init_ls", C.c_int), ("obj_scaling", C.c_int), ("verbose", C.c_int),
                ("polish", C.c_int), ("out_step_tol", C.c_double),
                ("bound_relax_factor", C.c_double), ("honor_original_bounds", C.c_int),
                ("dual_inf_tol", C.c_double), ("constr_viol_tol", C.c_double), ("compl_inf_tol", C.c_double),
                ("acceptable_tol", C.c_double), ("acceptable_dual_inf_tol", C.c_double),
                ("acceptable_constr_viol_tol", C.c_double), ("acceptable_compl_inf_tol", C.c_double),
                ("acceptable_iter", C.c_int), ("max_soc", C.c_int)]


class OrcSolveInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("kkt_error", C.c_double), ("mu", C.c_double),
                ("obj", C.c_double), ("constr_viol", C.c_double), ("dual_inf", C.c_double),
                ("compl_inf", C.c_double), ("n_regularised", C.c_int), ("n_backtracks", C.c_int),
                ("acceptable_restored_older", C.c_int), ("no_restart", C.c_int), ("n_soc_tried", C.c_int), ("n_soc_accepted", C.c_int)]


class OrcRunPre(C.Structure):
    _fields_ = [("nc", C.c_int), ("coef", C.c_double * MAX_COEF), ("state", C.c_double * 6),
                ("max_yaw_change", C.c_double), ("max_speed", C.c_double), ("target_speed", C.c_double),
                ("yaw_low", C.c_double), ("yaw_high", C.c_double)]


_lib = None
DP = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "mpc_oracle.c")
        if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
            build()
        L = C.CDLL(path)
        L.orc_polyeval.restype = C.c_double
        L.orc_polyeval.argtypes = [DP, C.c_int, C.c_double]
        L.orc_polyder.restype = C.c_double
        L.orc_polyder.argtypes = [DP, C.c_int, C.c_double]
        L.orc_mph2mps.restype = C.c_double
        L.orc_mph2mps.argtypes = [C.c_double]
        L.orc_normalize_angle.restype = C.c_double
        L.orc_normalize_angle.argtypes = [C.c_double]
        L.orc_speed_target.restype = C.c_double
        L.orc_speed_target.argtypes = [C.POINTER(OrcConfig), C.c_double, C.c_double]
        L.orc_yaw_change_speed_limit.restype = C.c_double
        L.orc_yaw_change_speed_limit.argtypes = [C.POINTER(OrcConfig), C.c_double, C.c_double]
        L.orc_compute_throttle.restype = C.c_double
        L.orc_compute_throttle.argtypes = [C.POINTER(OrcConfig)] + [C.c_double] * 4
        L.orc_orientation_change.restype = C.c_double
        L.orc_orientation_change.argtypes = [DP, C.c_int, C.c_double, C.c_double]
        L.orc_kkt_certificate.restype = C.c_double
        L.orc_config_load.argtypes = [C.c_char_p, C.POINTER(OrcConfig)]
        L.orc_polyfit.argtypes = [DP, DP, C.c_int, C.c_int, DP]
        L.orc_road_fit.argtypes = [DP, DP, C.c_int, C.c_int, C.c_double, DP, DP]
        L.orc_vehicle_move.argtypes = [C.POINTER(OrcConfig), DP, C.c_double]
        L.orc_global_to_vehicle.argtypes = [C.c_double] * 3 + [DP, DP, C.c_int]
        L.orc_fg_eval.argtypes = [C.POINTER(OrcConfig), DP, C.c_int, C.c_int, DP, DP, DP]
        L.orc_fg_grad.argtypes = [C.POINTER(OrcConfig), DP, C.c_int, C.c_int, DP, DP, DP, DP]
        L.orc_lag_hess.argtypes = [C.POINTER(OrcConfig), DP, C.c_int, C.c_int, DP, DP, C.c_double, DP, DP]
        L.orc_mpc_solve.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcSolveOptions), DP, DP, C.c_int,
                                    DP, DP, DP, DP, C.POINTER(OrcSolveInfo)]
        L.orc_mpc_run_pre.argtypes = [C.POINTER(OrcConfig), DP, DP, DP, C.c_int, C.POINTER(OrcRunPre)]
        L.orc_mpc_run_post.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcRunPre), C.c_double, DP, DP]
        L.orc_mpc_run.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcSolveOptions), DP, DP, DP, C.c_int, DP,
                                  DP, DP, C.POINTER(OrcRunPre), C.POINTER(OrcSolveInfo)]
        L.orc_kkt_certificate.argtypes = [C.POINTER(OrcConfig), DP, DP, C.c_int, DP, C.c_double, DP, DP, DP]
        _lib = L
    return _lib


def dptr(a):
    return a.ctypes.data_as(DP) if a is not None else None


def arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def load_config(name="config-stable.json", **overrides):
    cfg = OrcConfig()
    path = name if os.path.isabs(name) else os.path.join(GOLDEN, name)
    rc = lib().orc_config_load(path.encode(), C.byref(cfg))
    assert rc == 0, "config load failed: %s" % path
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def default_options(**kw):
    o = OrcSolveOptions()
    lib().orc_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def nvars(N):
    return 8 * N - 2


def fg_eval(cfg, coef, vars_, xi=None, branch_mode=0):
    coef = arr(coef); vars_ = arr(vars_)
    fg = np.zeros(1 + 6 * cfg.N)
    xi_a = arr(xi) if xi is not None else None
    lib().orc_fg_eval(C.byref(cfg), dptr(coef), len(coef), branch_mode, dptr(xi_a), dptr(vars_), dptr(fg))
    return fg


def fg_grad(cfg, coef, vars_, xi=None, branch_mode=0):
    coef = arr(coef); vars_ = arr(vars_)
    n = nvars(cfg.N); m = 6 * cfg.N
    g = np.zeros(n); J = np.zeros((m, n))
    xi_a = arr(xi) if xi is not None else None
    lib().orc_fg_grad(C.byref(cfg), dptr(coef), len(coef), branch_mode, dptr(xi_a), dptr(vars_), dptr(g), dptr(J))
    return g, J


def lag_hess(cfg, coef, vars_, obj_factor, lam, xi=None, branch_mode=0):
    coef = arr(coef); vars_ = arr(vars_); lam = arr(lam)
    n = nvars(cfg.N)
    H = np.zeros((n, n))
    xi_a = arr(xi) if xi is not None else None
    lib().orc_lag_hess(C.byref(cfg), dptr(coef), len(coef), branch_mode, dptr(xi_a), dptr(vars_),
                       obj_factor, dptr(lam), dptr(H))
    return H


def mpc_solve(cfg, state, coef, opt=None, want_sol=False):
    """-> (status, out9, traj_x, traj_y, info[, sol])"""
    state = arr(state); coef = arr(coef)
    out9 = np.zeros(9); tx = np.zeros(cfg.N); ty = np.zeros(cfg.N)
    sol = np.zeros(nvars(cfg.N))
    info = OrcSolveInfo()
    opt = opt if opt is not None else default_options()
    st = lib().orc_mpc_solve(C.byref(cfg), C.byref(opt), dptr(state), dptr(coef), len(coef), dptr(out9),
                             dptr(tx), dptr(ty), dptr(sol), C.byref(info))
    if want_sol:
        return st, out9, tx, ty, info, sol
    return st, out9, tx, ty, info


def run_pre(cfg, pose, ptsx, ptsy):
    """-> (OrcRunPre, ptsx_vehicle, ptsy_vehicle)"""
    pose = arr(pose); px = arr(ptsx).copy(); py = arr(ptsy).copy()
    pre = OrcRunPre()
    lib().orc_mpc_run_pre(C.byref(cfg), dptr(pose), dptr(px), dptr(py), len(px), C.byref(pre))
    return pre, px, py


def run_post(cfg, pre, v0, result9):
    r = arr(result9); out8 = np.zeros(8)
    lib().orc_mpc_run_post(C.byref(cfg), C.byref(pre), v0, dptr(r), dptr(out8))
    return out8


def mpc_run(cfg, pose, ptsx, ptsy, opt=None):
    """-> (status, out8, traj_x, traj_y, pre, info); cfg.yaw_low/high are mutated like the reference."""
    pose = arr(pose); px = arr(ptsx).copy(); py = arr(ptsy).copy()
    out8 = np.zeros(8); tx = np.zeros(cfg.N); ty = np.zeros(cfg.N)
    pre = OrcRunPre(); info = OrcSolveInfo()
    opt = opt if opt is not None else default_options()
    st = lib().orc_mpc_run(C.byref(cfg), C.byref(opt), dptr(pose), dptr(px), dptr(py), len(px), dptr(out8),
                           dptr(tx), dptr(ty), C.byref(pre), C.byref(info))
    return st, out8, tx, ty, pre, info


def kkt_certificate(cfg, state, coef, vars_, active_tol=1e-6):
    state = arr(state); coef = arr(coef); vars_ = arr(vars_)
    s = C.c_double(); p = C.c_double(); b = C.c_double()
    tot = lib().orc_kkt_certificate(C.byref(cfg), dptr(state), dptr(coef), len(coef), dptr(vars_),
                                    active_tol, C.byref(s), C.byref(p), C.byref(b))
    return tot, s.value, p.value, b.value


def vehicle_move(cfg, pose, dt):
    """Vehicle::move (Vehicle.cpp:145-168) on pose = {x,y,psi,v,steering,acceleration}."""
    p = arr(pose).copy()
    lib().orc_vehicle_move(C.byref(cfg), dptr(p), float(dt))
    return p


def compute_throttle(cfg, accel, target):
    return lib().orc_compute_throttle(C.byref(cfg), float(accel), float(target), cfg.max_acceleration, cfg.max_deceleration)


def telemetry_handler(cfg, tel6, ptsx, ptsy, extra=0.0, opt=None):
    """The onMessage handler around run(), src/mpc_main.cpp:126-174: tel6 = x, y, psi, speed[mph], steering_angle,
    previous throttle command -> (status, steer_value, throttle_value, out8)."""
    x, y, psi, mph, sa, thr = [float(t) for t in tel6]
    psi = lib().orc_normalize_angle(psi)
    v = mph * 1609.34 / 3600.0
    pose = np.array([x, y, psi, v, -sa, (thr - v / 50.0) * 6.0])
    if cfg.latency:
        pose = vehicle_move(cfg, pose, cfg.lookahead + extra)
    st, out8, tx, ty, pre, info = mpc_run(cfg, pose, ptsx, ptsy, opt)
    return st, -out8[4], compute_throttle(cfg, out8[5], out8[3]), out8


def solve_chunk(job):
    """Worker of bench.py's all-cores cpu_baseline leg (one process per core, one oracle thread each): solves the
    instances of `job` = (config name, overrides, state [6,n], coeffs [5,n], yaw_lo [n], yaw_hi [n], weights [12,n] or None)
    and returns how many it solved."""
    name, over, state, coeffs, ylo, yhi, w = job
    cfg = load_config(name, **over)
    n = state.shape[1]
    for i in range(n):
        cfg.yaw_low, cfg.yaw_high = float(ylo[i]), float(yhi[i])
        if w is not None:
            for q in range(12):
                cfg.weights[q] = float(w[q, i])
        mpc_solve(cfg, state[:, i], coeffs[:, i])
    return n


def solve_chunk_full(job):
    """Worker of tests/test_soak.py: like solve_chunk, but returns the oracle's results (out [9,n], traj [2N,n], status, iterations)."""
    name, over, state, coeffs, ylo, yhi, w = job
    cfg = load_config(name, **over)
    n = state.shape[1]
    out = np.zeros((9, n)); traj = np.zeros((2 * cfg.N, n)); status = np.zeros(n, dtype=np.int32); iters = np.zeros(n, dtype=np.int32)
    for i in range(n):
        cfg.yaw_low, cfg.yaw_high = float(ylo[i]), float(yhi[i])
        if w is not None:
            for q in range(12):
                cfg.weights[q] = float(w[q, i])
        st, o9, tx, ty, info = mpc_solve(cfg, state[:, i], coeffs[:, i])
        out[:, i] = o9; traj[:cfg.N, i] = tx; traj[cfg.N:, i] = ty; status[i] = st; iters[i] = info.iterations
    return out, traj, status, iters


def run_chunk_full(job):
    """Worker of tests/test_soak.py: the oracle's MPC::run() on poses + waypoints -> (status [n], out8 [8,n])."""
    name, over, pose, ptsx, ptsy = job
    n = pose.shape[1]
    out8 = np.zeros((8, n)); status = np.zeros(n, dtype=np.int32)
    for i in range(n):
        cfg = load_config(name, **over)          # run() mutates the yaw bounds of its Config, like the reference
        st, o8, _, _, _, _ = mpc_run(cfg, pose[:, i], ptsx[:, i], ptsy[:, i])
        out8[:, i] = o8; status[i] = st
    return status, out8


def rollout_chunk_full(job):
    """Worker of the closed-loop tests: `steps` oracle solves per car, each fed with the step-1 state of the one before
    (src/test.cpp:79-111) -> (worst status [n], hist [steps,9,n], status of every solve [steps,n])."""
    name, over, state, coeffs, ylo, yhi, steps = job
    cfg = load_config(name, **over)
    n = state.shape[1]
    hist = np.zeros((steps, 9, n)); worst = np.zeros(n, dtype=np.int32); every = np.zeros((steps, n), dtype=np.int32)
    for i in range(n):
        cfg.yaw_low, cfg.yaw_high = float(ylo[i]), float(yhi[i])
        s = list(state[:, i])
        for t in range(steps):
            st, o9, _, _, _ = mpc_solve(cfg, s, coeffs[:, i])
            hist[t, :, i] = o9; worst[i] = max(worst[i], st); every[t, i] = st
            s = list(o9[:6])
    return worst, hist, every
