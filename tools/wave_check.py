#!/usr/bin/env python3
"""GPU tool: the one-instance-per-wavefront kernel (MPC_WAVE_MAX_BATCH) against the lane-per-instance kernel on the same instances:
status, iterations, outputs; single-solve latency of both.   python tools/wave_check.py"""
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
for config, over, B in (("config-fast.json", {}, 256), ("config-stable.json", dict(N=25, dt=0.05), 64)):
    p = pkg.params_from_json(os.path.join(gd, config), **over); p.f64_f32_start = 0
    b = pkg.scenarios.lake_track_batch(B, p, wp, stream=3, filtered="survey")
    ins = (t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]))
    res = {}
    for mode in ("lane", "wave"):
        if mode == "wave": os.environ["MPC_WAVE_MAX_BATCH"] = "1024"; os.environ["MPC_WAVE_WHOLE_MAX"] = "0"      # (lanes per instance by the horizon: 16 / 32)
        else: os.environ["MPC_WAVE_MAX_BATCH"] = "0"
        with pkg.BatchedMPC(p, B, device=0) as mpc:
            r = mpc.solve_torch(*ins, want_traj=True); torch.cuda.synchronize()
            res[mode] = {k: v.cpu().numpy() for k, v in r.items()}
        with pkg.BatchedMPC(p, 1, device=0) as mpc:
            one = tuple(x[..., :1].contiguous() for x in ins)
            for _ in range(5): mpc.solve_torch(*one, want_traj=True); torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); mpc.solve_torch(*one, want_traj=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[mode]["b1_ms"] = 1e3 * float(np.median(ts))
    a, c = res["lane"], res["wave"]
    ok = (a["status"] == 0) & (c["status"] == 0)
    print(json.dumps({"config": config, "N": p.N, "B": B, "status_equal": bool(np.array_equal(a["status"], c["status"])), "iters_equal": float((a["iters"] == c["iters"]).mean()),
                      "bitwise": bool(np.array_equal(a["out"], c["out"]) and np.array_equal(a["traj"], c["traj"])),
                      "max_abs_out_diff_converged": float(np.abs(a["out"][:8] - c["out"][:8])[:, ok].max()) if ok.any() else None,
                      "b1_ms_lane": a["b1_ms"], "b1_ms_wave": c["b1_ms"], "status_counts": np.bincount(a["status"], minlength=4).tolist()}))
