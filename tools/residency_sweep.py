"""Experiment: launch time of the 65 536-instance workload versus resident waves per CU and gains placement."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
params = pkg.params_from_json(root + '/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(root + '/tests/golden/lake_track_waypoints.csv')
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for B in (65536, 4096):
    b = pkg.scenarios.lake_track_batch(B, params, wp)
    st, cf, yl, yh = t(b['state']), t(b['coeffs']), t(b['yaw_lo']), t(b['yaw_hi'])
    for w in (1, 2, 3, 4, 5):
        for g in ((1, 0) if w <= 2 else (0,)):
            os.environ['MPC_WAVES_PER_CU'] = str(w); os.environ['MPC_GAINS_IN_LDS'] = str(g)
            mpc = pkg.BatchedMPC(params, B, device=0)
            outs = mpc.alloc_outputs(B, dev, want_traj=True)
            ts = []
            for rep in range(6):
                torch.cuda.synchronize(); t0 = time.time()
                mpc.solve_torch(st, cf, yl, yh, outputs=outs)
                torch.cuda.synchronize(); ts.append(time.time() - t0)
            s = mpc.stats()
            print('B %6d waves/CU %d gains_in_lds %d: best %.3f ms median %.3f ms  -> %.3g solves/s  (succ %d, iters %.2f max %d)' % (
                B, w, g, min(ts) * 1e3, np.median(ts) * 1e3, B / min(ts), s.n_success, s.iter_sum / B, s.iter_max), flush=True)
            mpc.close()
