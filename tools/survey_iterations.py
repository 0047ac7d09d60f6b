"""CPU tool: iteration counts and statuses of the headline population (filtered and SURVEY 8d's) by the CPU build of the device solver
(tests/host_twin), 8 processes: writes /tmp/w/iters_{filtered,survey}.npz, which tools/soc_check.py reads."""
import os
os.makedirs("/tmp/w", exist_ok=True)
import os, sys, ctypes as C, numpy as np, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as G
pkg = G.load_package()
from helpers import twin_solve
import multiprocessing as mp
gd = "/root/repo/tests/golden"
def work(job):
    import ctypes as C
    sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
    import __graft_entry__ as G
    pkg = G.load_package()
    from helpers import twin_solve
    twin = C.CDLL("/root/repo/tests/host_twin/libhost_twin.so")
    params = pkg.params_from_json(os.path.join(gd, "config-fast.json"))
    r = twin_solve(twin, params, job, want_traj=False)
    return r["iters"], r["status"]
if __name__ == "__main__":
    wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
    params = pkg.params_from_json(os.path.join(gd, "config-fast.json"))
    for pop in ("filtered", "survey"):
        B = 65536
        b = pkg.scenarios.lake_track_batch(B, params, wp, stream=3, filtered=(True if pop == "filtered" else "survey"))
        chunks = [{k: np.ascontiguousarray(b[k][..., i:i + 2048]) for k in ("state", "coeffs", "yaw_lo", "yaw_hi")} for i in range(0, B, 2048)]
        t0 = time.time()
        with mp.Pool(8) as pool:
            res = pool.map(work, chunks)
        it = np.concatenate([r[0] for r in res]); st = np.concatenate([r[1] for r in res])
        print(pop, "twin time", time.time() - t0, "mean iters", it.mean(), "max", it.max(), np.bincount(st))
        np.savez("/tmp/w/iters_%s.npz" % pop, iters=it, status=st, state=b["state"], coeffs=b["coeffs"], yaw_lo=b["yaw_lo"], yaw_hi=b["yaw_hi"])
