"""One small mixed-precision (f32_finish) solve on the device with stderr visible: B instances, statuses and the difference to fp64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
B = int(os.environ.get("B", 64))
params = pkg.params_from_json(ROOT + '/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(ROOT + '/tests/golden/lake_track_waypoints.csv')
b = pkg.scenarios.lake_track_batch(B, params, wp, seed=53)
dev = torch.device('cuda:0')
res = {}
for name, prec, dt in (("f64", pkg.PRECISION_F64, torch.float64), ("mixed", pkg.PRECISION_F32, torch.float32)):
    p = params.copy(); p.precision = prec
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt)
    print("create", name, flush=True)
    with pkg.BatchedMPC(p, B, device=0) as mpc:
        print("launch", name, flush=True)
        r = mpc.solve_torch(t(b['state']), t(b['coeffs']), t(b['yaw_lo']), t(b['yaw_hi']), want_traj=True)
        torch.cuda.synchronize()
        print("done", name, flush=True)
        res[name] = {k: v.cpu().numpy() for k, v in r.items()}
    print(name, "status", np.bincount(res[name]["status"], minlength=6), "iters mean", res[name]["iters"].mean(), flush=True)
d = np.abs(res["mixed"]["out"].astype(np.float64) - res["f64"]["out"])
print("max diff rows", d.max(1))
