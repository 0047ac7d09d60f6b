#!/usr/bin/env python3
"""f64_f32_start against the single-phase solve on the SAME instances of SURVEY 8d's unfiltered populations (the headline's 65 536
at N = 10, configs[3]'s share of 32 768 at N = 25) and of the filtered headline batch: status by status, outputs at 1e-6.
GPU; prints one JSON line per population."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
gd = os.path.join(ROOT, "tests", "golden")
wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for name, config, over, B, filt in (("headline, SURVEY's rejection only", "config-fast.json", {}, 65536, "survey"),
                                    ("configs[3] share, SURVEY's rejection only", "config-stable.json", dict(N=25, dt=0.05), 32768, "survey"),
                                    ("headline, filtered", "config-fast.json", {}, 65536, True)):
    res = []
    for start in (0, 1):
        p = pkg.params_from_json(os.path.join(gd, config), **over); p.f64_f32_start = start
        b = pkg.scenarios.lake_track_batch(B, p, wp, stream=3, filtered=filt)
        with pkg.BatchedMPC(p, B, device=0) as mpc:
            r = mpc.solve_torch(t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]), want_traj=True)
            torch.cuda.synchronize()
            res.append({k: v.cpu().numpy() for k, v in r.items() if v is not None})
    a, c = res
    ok = (a["status"] == 0) & (c["status"] == 0)
    d = np.abs(a["out"][:8] - c["out"][:8])[:, ok]
    dt_ = np.abs(a["traj"] - c["traj"])[:, ok].max(0)
    print(json.dumps({"population": name, "instances": B, "status_single_phase": np.bincount(a["status"], minlength=5).tolist(),
                      "status_f32_start": np.bincount(c["status"], minlength=5).tolist(), "status_differs": int((a["status"] != c["status"]).sum()),
                      "both_converged": int(ok.sum()), "beyond_1e-6 (delta0, a0) or 1e-5 (state, trajectory)": int(((d[6] > 1e-6) | (d[7] > 1e-6) | (d[:6].max(0) > 1e-5) | (dt_ > 1e-5)).sum()),
                      "max": {"delta0": float(d[6].max()), "a0": float(d[7].max()), "state": float(d[:6].max()), "trajectory_m": float(dt_.max())},
                      "mean_iterations": [float(a["iters"].mean()), float(c["iters"].mean())], "max_iterations": [int(a["iters"].max()), int(c["iters"].max())]}))
