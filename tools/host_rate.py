"""How fast can the host enqueue steps?  (enqueue-only timing of BatchedMPC.solve_torch on small batches.)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
params = pkg.params_from_json(ROOT + '/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(ROOT + '/tests/golden/lake_track_waypoints.csv')
dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for B, nfl in ((256, 4), (8192, 4), (32768, 4)):
    b = pkg.scenarios.lake_track_batch(B, params, wp)
    st, cf, yl, yh = t(b['state']), t(b['coeffs']), t(b['yaw_lo']), t(b['yaw_hi'])
    mpcs = [pkg.BatchedMPC(params, B, device=0) for _ in range(nfl)]
    outs = [m.alloc_outputs(B, dev, want_traj=True) for m in mpcs]
    streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(nfl)]
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % nfl]):
            mpcs[i % nfl].solve_torch(st, cf, yl, yh, outputs=outs[i % nfl])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B %6d inflight %d: enqueue %.1f us/step, total %.1f us/step" % (B, nfl, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n), flush=True)
    for m in mpcs: m.close()
