import os, sys, numpy as np, threading
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tests") else ".")
sys.path.insert(0, "tests")
import torch
import __graft_entry__ as G
pkg = G.load_package()
gd = "tests/golden"
wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
dev = torch.device("cuda:0")
def run(pool, params, b, w=None, nh=3, rounds=6, dtype=torch.float64):
    if pool: os.environ["MPC_TILE_POOL"] = "1"
    else: os.environ.pop("MPC_TILE_POOL", None)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dtype)
    B = b["state"].shape[1]
    ins = (t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]))
    wt = t(w) if w is not None else None
    hs = [pkg.BatchedMPC(params, B, device=0) for _ in range(nh)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nh)]
    res = []
    for r in range(rounds):
        outs = []
        for h, s in zip(hs, streams):
            with torch.cuda.stream(s):
                outs.append(h.solve_torch(*ins, weights=wt, want_traj=True))
        torch.cuda.synchronize()
        res.append([{k: v.cpu().numpy() for k, v in o.items()} for o in outs])
    for h in hs: h.close()
    return res
for name, B, prec, dtype, N, dt, sweep in (("head", 65536, pkg.PRECISION_F64, torch.float64, 10, 0.1, False), ("f32 sweep", 65536, pkg.PRECISION_F32, torch.float32, 10, 0.1, True), ("N25", 16384, pkg.PRECISION_F64, torch.float64, 25, 0.05, False), ("ragged", 16384 + 11, pkg.PRECISION_F64, torch.float64, 10, 0.1, True)):
    params = pkg.params_from_json(os.path.join(gd, "config-fast.json"), N=N, dt=dt); params.precision = prec
    b = pkg.scenarios.lake_track_batch(B, params, wp, seed=7)
    w = pkg.scenarios.weight_sweep(B, params, seed=8) if sweep else None
    ref = run(False, params, b, w, nh=1, rounds=1, dtype=dtype)[0][0]
    got = run(True, params, b, w, nh=3, rounds=6, dtype=dtype)
    bad = 0
    for rnd in got:
        for o in rnd:
            for k in ("out", "traj", "status", "iters"):
                if not np.array_equal(o[k], ref[k]): bad += 1
    print(name, "pooled launches differing from the plain one:", bad, "of", 6 * 3 * 4, flush=True)
