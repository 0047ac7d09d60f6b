"""Throughput with 1, 2 and 3 batches in flight (separate handles and streams): does the next batch fill the SIMDs that a
launch frees in its tail?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package()
params = pkg.params_from_json(ROOT + '/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(ROOT + '/tests/golden/lake_track_waypoints.csv')
B = int(os.environ.get("B", 65536)); K = int(os.environ.get("K", 24))
b = pkg.scenarios.lake_track_batch(B, params, wp)
dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
st, cf, yl, yh = t(b['state']), t(b['coeffs']), t(b['yaw_lo']), t(b['yaw_hi'])
for nfl in (1, 2, 3, 1, 2):
    hs = [pkg.BatchedMPC(params, B, device=0) for _ in range(nfl)]
    outs = [h.alloc_outputs(B, dev, want_traj=True) for h in hs]
    ss = [torch.cuda.Stream(device=dev, priority=int(os.environ.get('PRIO', -1))) for _ in range(nfl)]
    for i in range(nfl):
        hs[i].solve_torch(st, cf, yl, yh, outputs=outs[i], stream=ss[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        j = i % nfl
        hs[j].solve_torch(st, cf, yl, yh, outputs=outs[j], stream=ss[j])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(outs[0]["out"], o["out"]) for o in outs)
    print("in flight %d: %.3f ms per batch, %.3g solves/s, outputs identical across handles: %s" % (nfl, dt / K * 1e3, B * K / dt, same), flush=True)
    for h in hs: h.close()
