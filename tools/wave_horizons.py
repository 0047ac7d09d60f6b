import json, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import __graft_entry__ as G
pkg = G.load_package()
gd = "/root/repo/tests/golden"
wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
for N, dt in ((17, 0.07), (18, 0.07), (33, 0.04), (34, 0.04), (40, 0.03), (64, 0.02)):
    p = pkg.params_from_json(os.path.join(gd, "config-stable.json"), N=N, dt=dt); p.f64_f32_start = 0
    B = 96
    b = pkg.scenarios.lake_track_batch(B, p, wp, stream=3, filtered="survey")
    ins = (t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]))
    res = {}
    for mode in ("lane", "wave"):
        os.environ["MPC_WAVE_MAX_BATCH"] = "0" if mode == "lane" else "100000"
        with pkg.BatchedMPC(p, B, device=0) as mpc:
            r = mpc.solve_torch(*ins, want_traj=True); torch.cuda.synchronize()
            res[mode] = {k: v.cpu().numpy() for k, v in r.items()}
    a, c = res["lane"], res["wave"]
    print(N, "bitwise", all(np.array_equal(a[k], c[k], equal_nan=True) for k in ("status", "iters", "out", "traj")), np.bincount(a["status"], minlength=4).tolist(), "max iters", int(a["iters"].max()))
