import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg=G.load_package()
G.smoke()
params = pkg.params_from_json(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'/tests/golden/config-fast.json')
wp = pkg.scenarios.load_waypoints(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'/tests/golden/lake_track_waypoints.csv')
for B in (4096, 65536):
    b = pkg.scenarios.lake_track_batch(B, params, wp)
    dev=torch.device('cuda:0'); t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    st,cf,yl,yh = t(b['state']),t(b['coeffs']),t(b['yaw_lo']),t(b['yaw_hi'])
    mpc = pkg.BatchedMPC(params, B, device=0)
    outs = mpc.alloc_outputs(B, dev, want_traj=True)
    for rep in range(5):
        torch.cuda.synchronize(); t0=time.time()
        mpc.solve_torch(st,cf,yl,yh,outputs=outs)
        torch.cuda.synchronize(); dt=time.time()-t0
        s=mpc.stats()
        print('B',B,'rep',rep,'wall %.3f ms'%(dt*1e3),'kernel %.3f ms'%s.kernel_ms,'solves/s %.3g'%(B/dt),'succ',s.n_success,'iters mean %.2f max %d'%(s.iter_sum/B,s.iter_max), flush=True)
    mpc.close()
