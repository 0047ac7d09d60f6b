#!/usr/bin/env python3
"""GPU tool: one launch of B instances, lane-per-instance kernel against the one-instance-per-wavefront kernel (median of 30)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as G
pkg = G.load_package()
gd = os.path.join(ROOT, "tests", "golden")
wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
p = pkg.params_from_json(os.path.join(gd, "config-fast.json"))
b = pkg.scenarios.lake_track_batch(8192, p, wp, stream=3, filtered=True)
full = (t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]))
for B in (1, 8, 64, 256, 1024, 2048, 4096, 8192):
    ins = tuple(x[..., :B].contiguous() for x in full)
    row = {"B": B}
    for mode in ("lane_fp64", "wave64", "wave16"):
        q = p.copy(); q.f64_f32_start = 0
        os.environ["MPC_WAVE_MAX_BATCH"] = "0" if mode == "lane_fp64" else "1000000"
        os.environ["MPC_WAVE_LPI"] = "16" if mode == "wave16" else "64"
        with pkg.BatchedMPC(q, B, device=0) as mpc:
            for _ in range(5): mpc.solve_torch(*ins, want_traj=True); torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); mpc.solve_torch(*ins, want_traj=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row[mode + "_ms"] = round(1e3 * float(np.median(ts)), 3)
    print(json.dumps(row), flush=True)
