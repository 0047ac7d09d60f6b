#!/usr/bin/env python3
"""The shipped MPC_PRECISION_F32 mode against the fp64 solve on the SAME instances of configs[4]'s share drawn with SURVEY 8d's
rejection only (131 072 instances, per-instance weights incl. velocity weight 0): status by status, and the outputs where both
converge.  GPU; prints one JSON line."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device("cuda:0")
res = {}
for prec, dt_ in ((pkg.PRECISION_F64, torch.float64), (pkg.PRECISION_F32, torch.float32)):
    p = pkg.params_from_json(os.path.join(gd, "config-fast.json")); p.precision = prec
    b = pkg.scenarios.lake_track_batch(B, p, wp, stream=3, filtered="survey")
    w = pkg.scenarios.weight_sweep(B, p, seed=1234, velocity_weights=(0.0, 1.0, 100.0))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt_)
    with pkg.BatchedMPC(p, B, device=0) as mpc:
        r = mpc.solve_torch(t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]), weights=t(w), want_traj=False)
        torch.cuda.synchronize()
        res[prec] = {k: v.cpu().numpy() for k, v in r.items() if v is not None}
# the fp64 solve on the inputs as an fp32 array holds them: what the rounding of the inputs alone does to the hard instances
p = pkg.params_from_json(os.path.join(gd, "config-fast.json"))
b = pkg.scenarios.lake_track_batch(B, p, wp, stream=3, filtered="survey")
w = pkg.scenarios.weight_sweep(B, p, seed=1234, velocity_weights=(0.0, 1.0, 100.0))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.float32).astype(np.float64)).to(dev)
with pkg.BatchedMPC(p, B, device=0) as mpc:
    r = mpc.solve_torch(t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]), weights=t(w), want_traj=False)
    torch.cuda.synchronize()
    rr = {k: v.cpu().numpy() for k, v in r.items() if v is not None}
a, c = res[pkg.PRECISION_F64], res[pkg.PRECISION_F32]
ok2 = (a["status"] == 0) & (rr["status"] == 0)
d2 = np.abs(a["out"][:8] - rr["out"][:8])[:, ok2]
okc = (c["status"] == 0) & (rr["status"] == 0)
d3 = np.abs(c["out"][:8].astype(np.float64) - rr["out"][:8])[:, okc]
rounding = {"what": "fp64 solve on float32-rounded inputs against the fp64 solve on the original inputs", "status_differs": int((a["status"] != rr["status"]).sum()),
            "beyond_1e-3": int(((d2[6] > 1e-3) | (d2[7] > 1e-3) | (d2[:6].max(0) > 1e-3)).sum()),
            "f32_mode_vs_fp64_on_the_SAME_rounded_inputs": {"status_differs": int((c["status"] != rr["status"]).sum()),
                                                            "beyond_1e-3": int(((d3[6] > 1e-3) | (d3[7] > 1e-3) | (d3[:6].max(0) > 1e-3)).sum())}}
ok = (a["status"] == 0) & (c["status"] == 0)
d = np.abs(a["out"][:8] - c["out"][:8].astype(np.float64))[:, ok]
print(json.dumps({"instances": B, "status_fp64": np.bincount(a["status"], minlength=5).tolist(), "status_f32_mode": np.bincount(c["status"], minlength=5).tolist(),
                  "status_differs": int((a["status"] != c["status"]).sum()), "both_converged": int(ok.sum()),
                  "beyond_1e-3": {"delta0": int((d[6] > 1e-3).sum()), "a0": int((d[7] > 1e-3).sum()), "state": int((d[:6].max(0) > 1e-3).sum())},
                  "max": {"delta0": float(d[6].max()), "a0": float(d[7].max()), "state": float(d[:6].max())},
                  "p999": {"delta0": float(np.quantile(d[6], 0.999)), "a0": float(np.quantile(d[7], 0.999)), "state": float(np.quantile(d[:6].max(0), 0.999))},
                  "mean_iterations": [float(a["iters"].mean()), float(c["iters"].mean())], "input_rounding_alone": rounding}))
