"""Differences between two settings of one environment variable on one batch:  python tools/ab_env.py VAR A B [N dt batch]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
var, va, vb = sys.argv[1:4]
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dt = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
B = int(sys.argv[6]) if len(sys.argv) > 6 else 16384 + 11
params = pkg.params_from_json(ROOT + '/tests/golden/config-fast.json', N=N, dt=dt)
wp = pkg.scenarios.load_waypoints(ROOT + '/tests/golden/lake_track_waypoints.csv')
dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
b = pkg.scenarios.lake_track_batch(B, params, wp, seed=78)
res = {}
for v in (va, vb):
    os.environ[var] = v
    mpc = pkg.BatchedMPC(params, B, device=0)
    r = mpc.solve_torch(t(b['state']), t(b['coeffs']), t(b['yaw_lo']), t(b['yaw_hi']), want_traj=True)
    torch.cuda.synchronize()
    res[v] = {k: x.cpu().numpy() for k, x in r.items() if x is not None}
    s = mpc.stats(); print(var, v, "succ", s.n_success, "iters mean %.3f max %d" % (s.iter_sum / B, s.iter_max), "kernel %.3f ms" % s.kernel_ms)
    mpc.close()
d = np.abs(res[va]["out"] - res[vb]["out"])
bad = np.where((d.max(0) > 0) | (res[va]["iters"] != res[vb]["iters"]) | (res[va]["status"] != res[vb]["status"]))[0]
print("instances differing:", len(bad), "of", B, "max abs diff per row:", d.max(1))
print("iters of differing (a):", res[va]["iters"][bad][:20], "(b):", res[vb]["iters"][bad][:20])
print("status (a):", np.bincount(res[va]["status"], minlength=5), "(b):", np.bincount(res[vb]["status"], minlength=5))
