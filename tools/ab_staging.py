"""A/B: LDS-DMA staging on/off (MPC_STAGING), 65 536- and 4 096-instance workloads, interleaved in one process."""
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
    hs = {}
    for stg in (1, 0):
        os.environ['MPC_STAGING'] = str(stg)
        hs[stg] = (pkg.BatchedMPC(params, B, device=0),)
        hs[stg] += (hs[stg][0].alloc_outputs(B, dev, want_traj=True),)
    res = {1: [], 0: []}
    for rep in range(8):
        for stg in (1, 0):
            mpc, outs = hs[stg]
            torch.cuda.synchronize(); t0 = time.time()
            mpc.solve_torch(st, cf, yl, yh, outputs=outs)
            torch.cuda.synchronize(); res[stg].append(time.time() - t0)
    same = torch.equal(hs[1][1]['out'], hs[0][1]['out']) and torch.equal(hs[1][1]['traj'], hs[0][1]['traj'])
    for stg in (1, 0):
        s = hs[stg][0].stats()
        print('B %6d staging %d: best %.3f ms median %.3f ms -> %.3g solves/s (succ %d, iters %.2f max %d)' % (
            B, stg, min(res[stg]) * 1e3, np.median(res[stg]) * 1e3, B / min(res[stg]), s.n_success, s.iter_sum / B, s.iter_max), flush=True)
    print('   outputs bitwise identical between the two variants:', same, flush=True)
    for stg in (1, 0):
        hs[stg][0].close()
