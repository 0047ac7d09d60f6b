"""Differences between the staged and the plain kernel (MPC_STAGING=1/0) on one batch: how many instances differ and by how much."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
params = pkg.params_from_json(ROOT + '/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(ROOT + '/tests/golden/lake_track_waypoints.csv')
dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
B = 12288 + 37
b = pkg.scenarios.lake_track_batch(B, params, wp, seed=77)
res = {}
for stg in ("1", "0", "1"):
    os.environ["MPC_STAGING"] = stg
    mpc = pkg.BatchedMPC(params, B, device=0)
    r = mpc.solve_torch(t(b['state']), t(b['coeffs']), t(b['yaw_lo']), t(b['yaw_hi']), want_traj=True)
    torch.cuda.synchronize()
    cur = {k: v.cpu().numpy() for k, v in r.items() if v is not None}
    if stg in res:
        print("staged run twice identical:", all(np.array_equal(cur[k], res[stg][k]) for k in cur))
    res[stg] = cur
    mpc.close()
d = np.abs(res["1"]["out"] - res["0"]["out"])
bad = np.where(d.max(0) > 0)[0]
print("instances differing:", len(bad), "of", B, "max abs diff per row:", d.max(1))
print("iters differ:", int((res["1"]["iters"] != res["0"]["iters"]).sum()), "status differ:", int((res["1"]["status"] != res["0"]["status"]).sum()))
print("first differing instances:", bad[:10], "lanes:", bad[:10] % 64)
