"""CPU tool: iteration counts of the CPU build of the device solver (tests/host_twin) against the oracle on SURVEY's populations.
   python tools/iter_eq.py tests/host_twin/libhost_twin.so 2048 [initial_state_rows]"""
import os, sys, ctypes as C, numpy as np, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as G
pkg = G.load_package()
import oracle_lib as O
from helpers import twin_solve
import multiprocessing as mp
gd = "/root/repo/tests/golden"
def orc(job):
    name, over, b, idx, w = job
    from helpers import oracle_solve_batch
    import oracle_lib as O
    return oracle_solve_batch(O.load_config(name, **over), b, idx, weights=w)
if __name__ == "__main__":
    wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
    twin = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else "/root/repo/tests/host_twin/libhost_twin.so")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    for name, over, pop, sweep in (("config-fast.json", {}, True, False), ("config-fast.json", {}, "survey", False), ("config-stable.json", dict(N=25, dt=0.05), "survey", False), ("config-fast.json", {}, "survey", True)):
        params = pkg.params_from_json(os.path.join(gd, name), **over)
        params.f64_f32_start = 0; params.initial_state_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        b = pkg.scenarios.lake_track_batch(n, params, wp, stream=3, filtered=pop)
        w = pkg.scenarios.weight_sweep(n, params, seed=1234, velocity_weights=(0.0, 1.0, 100.0)) if sweep else None
        r = twin_solve(twin, params, b, weights=w)
        chunks = np.array_split(np.arange(n), 8)
        with mp.Pool(8) as pool:
            parts = pool.map(orc, [(name, over, b, [int(i) for i in ch], w) for ch in chunks])
        ost = np.concatenate([p["status"] for p in parts]); oit = np.concatenate([p["iters"] for p in parts]); oo = np.concatenate([p["out"] for p in parts], axis=1)
        ok = (r["status"] == 0) & (ost == 0)
        same = (r["iters"] == oit) & (r["status"] == ost)
        d = np.abs(r["out"][6] - oo[6])[ok]
        print("%s N=%d pop=%s sweep=%s: same status %.4f, same iters %.4f (of converged %.4f), max |diters| %d, twin mean it %.3f oracle %.3f, dsteer max %.2e p99 %.2e" % (
            name, params.N, pop, sweep, (r["status"] == ost).mean(), same.mean(), (r["iters"][ok] == oit[ok]).mean(), np.abs(r["iters"][ok] - oit[ok]).max(), r["iters"].mean(), oit.mean(), d.max(), np.quantile(d, 0.99)))
        bad = np.where(~same)[0][:10]
        print("   first differing:", bad.tolist(), r["iters"][bad].tolist(), oit[bad].tolist())
