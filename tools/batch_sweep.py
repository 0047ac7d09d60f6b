"""Kernel time versus batch size (same instances, prefixes of one batch): separates the latency-bound regime (few waves)
from the bandwidth-bound one and shows where the workspace stops fitting the Infinity Cache."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
params = pkg.params_from_json(ROOT + '/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(ROOT + '/tests/golden/lake_track_waypoints.csv')
BMAX = int(os.environ.get("BMAX", 131072))
b = pkg.scenarios.lake_track_batch(BMAX, params, wp, seed=int(os.environ.get('SEED', pkg.scenarios.DEFAULT_SEED)))
dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for B in [int(x) for x in os.environ.get("BS", "1024,4096,8192,16384,32768,49152,65536,98304,131072").split(",")]:
    st, cf, yl, yh = t(b['state'][:, :B]), t(b['coeffs'][:, :B]), t(b['yaw_lo'][:B]), t(b['yaw_hi'][:B])
    mpc = pkg.BatchedMPC(params, B, device=0)
    outs = mpc.alloc_outputs(B, dev, want_traj=True)
    ks = []
    for rep in range(int(os.environ.get('REPS', 6))):
        mpc.solve_torch(st, cf, yl, yh, outputs=outs)
        if not os.environ.get('NOSYNC'): torch.cuda.synchronize()
        s = mpc.stats(); ks.append(s.kernel_ms)
    k = float(np.median(ks[1:]))
    print('B %6d kernel %.3f ms  %.3g solves/s  ns/solve %.1f  iters mean %.2f max %d succ %d' % (B, k, B / k * 1e3, k * 1e6 / B, s.iter_sum / B, s.iter_max, s.n_success), flush=True)
    mpc.close()
