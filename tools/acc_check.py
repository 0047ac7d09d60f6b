"""Acceptable-level termination: twin (device solver, CPU build) vs oracle on the hard instances of SURVEY's populations."""
import os, sys, time, json, ctypes as C
import multiprocessing as mp
import numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import __graft_entry__ as G
pkg = G.load_package()
from helpers import twin_solve
import oracle_lib as O
twin = C.CDLL(ROOT+"/tests/host_twin/libhost_twin.so")
golden = ROOT+"/tests/golden"
wp = pkg.scenarios.load_waypoints(golden+"/lake_track_waypoints.csv")

def oracle_job(job):
    name, over, st, cf, yl, yh, w, max_iter = job
    cfg = O.load_config(name, **over)
    opt = O.default_options(max_iter=max_iter)
    n = st.shape[1]
    res = []
    for i in range(n):
        cfg.yaw_low, cfg.yaw_high = float(yl[i]), float(yh[i])
        if w is not None:
            for q in range(12): cfg.weights[q] = float(w[q, i])
        s, o9, tx, ty, info = O.mpc_solve(cfg, st[:, i], cf[:, i], opt)
        res.append((s, info.iterations, info.acceptable_restored_older, info.no_restart, list(o9)))
    return res

def run(name, config, over, B, sweep=False, off=None):
    params = pkg.params_from_json(golden+"/"+config, **over)
    if off: params.acceptable_iter = 0
    b = pkg.scenarios.lake_track_batch(B, params, wp, stream=3, filtered="survey")
    w = pkg.scenarios.weight_sweep(B, params, seed=1234, velocity_weights=(0.0,1.0,100.0)) if sweep else None
    t=time.time(); r = twin_solve(twin, params, b, weights=w, want_traj=False); tt=time.time()-t
    it, st = r["iters"], r["status"]
    print(name, "twin %.1fs status %s mean it %.3f max %d" % (tt, np.bincount(st, minlength=7).tolist(), it.mean(), it.max()), flush=True)
    hard = np.where((it > 22) | (st != 0))[0]
    rnd = np.random.default_rng(1).choice(B, 256, replace=False)
    idx = np.unique(np.concatenate([hard, rnd]))
    chunks = np.array_split(idx, 32)
    jobs = [(config, over, b["state"][:, c].copy(), b["coeffs"][:, c].copy(), b["yaw_lo"][c].copy(), b["yaw_hi"][c].copy(), None if w is None else w[:, c].copy(), params.max_iter) for c in chunks]
    t=time.time()
    with mp.get_context("spawn").Pool(8) as pool:
        res = pool.map(oracle_job, jobs)
    flat = [x for ch in res for x in ch]
    ost = np.array([x[0] for x in flat]); oit = np.array([x[1] for x in flat]); older = np.array([x[2] for x in flat]); nr = np.array([x[3] for x in flat])
    oo = np.array([x[4] for x in flat]).T
    print("  oracle on %d instances %.0fs: status %s ; restored_older %d no_restart %d" % (len(idx), time.time()-t, np.bincount(ost, minlength=7).tolist(), older.sum(), nr.sum()))
    mism = np.where(ost != st[idx])[0]
    print("  status mismatches twin vs oracle:", len(mism), [(int(idx[j]), int(st[idx[j]]), int(ost[j]), int(it[idx[j]]), int(oit[j])) for j in mism[:12]])
    same = (ost == st[idx]) & ((ost == 0) | (ost == 6))
    d = np.abs(oo[:8] - r["out"][:8, idx])
    print("  max |d out| on same-status success/acceptable: %.3g (n=%d); iteration count differs on %d" % (d[:, same].max(), same.sum(), (oit != it[idx]).sum()))
    acc = np.where(st == 6)[0]
    print("  acceptable instances:", [(int(i), int(it[i])) for i in acc[:20]])
    return r

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "n10"
    off = len(sys.argv) > 2 and sys.argv[2] == "off"
    if which == "n10": run("N=10 survey 65536", "config-fast.json", {}, 65536, off=off)
    if which == "n25": run("N=25 survey 32768", "config-stable.json", dict(N=25, dt=0.05), 32768, off=off)
    if which == "w": run("sweep survey 65536", "config-fast.json", {}, 65536, sweep=True, off=off)
