"""CPU tool: the oracle with and without IPOPT's second-order correction (OrcSolveOptions.max_soc) on the hard instances of SURVEY's
N = 10 population and on a random sample.  Needs /tmp/w/iters_survey.npz (iteration counts of the population; written by an
earlier run of the CPU build) -- DESIGN.md section 3 quotes the output."""
import os, sys, numpy as np, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import multiprocessing as mp
def job(a):
    import oracle_lib as O
    st, cf, yl, yh, soc = a
    cfg = O.load_config("config-fast.json")
    opt = O.default_options(max_iter=200, max_soc=soc)
    res = []
    for i in range(st.shape[1]):
        cfg.yaw_low, cfg.yaw_high = float(yl[i]), float(yh[i])
        s, o9, tx, ty, info = O.mpc_solve(cfg, st[:, i], cf[:, i], opt)
        res.append((s, info.iterations, info.n_soc_tried, info.n_soc_accepted, info.n_backtracks, o9[6]))
    return res
if __name__ == "__main__":
    d = np.load("/tmp/w/iters_survey.npz")
    it, stt = d["iters"], d["status"]
    hard = np.where((it > 22) | (stt != 0))[0]
    rnd = np.random.default_rng(1).choice(len(it), 1024, replace=False)
    for name, idx in (("hard", hard), ("random", rnd)):
        out = {}
        for soc in (0, 4):
            chunks = np.array_split(idx, 16)
            with mp.Pool(8) as pool:
                parts = pool.map(job, [(d["state"][:, ch].copy(), d["coeffs"][:, ch].copy(), d["yaw_lo"][ch].copy(), d["yaw_hi"][ch].copy(), soc) for ch in chunks])
            r = np.array([x for p in parts for x in p], dtype=float)
            out[soc] = r
            print(name, "max_soc", soc, "n", len(idx), "status", np.bincount(r[:, 0].astype(int), minlength=7).tolist(), "iters mean %.2f sum %d max %d" % (r[:, 1].mean(), r[:, 1].sum(), r[:, 1].max()),
                  "soc tried %d accepted %d backtracks %d" % (r[:, 2].sum(), r[:, 3].sum(), r[:, 4].sum()))
        a, b = out[0], out[4]
        both = (a[:, 0] == 0) & (b[:, 0] == 0)
        print("   both converged", int(both.sum()), "max |dsteer| %.2e" % np.abs(a[both, 5] - b[both, 5]).max(), "n differing > 1e-6:", int((np.abs(a[both, 5] - b[both, 5]) > 1e-6).sum()),
              "status changed", int((a[:, 0] != b[:, 0]).sum()), "iters changed", int((a[:, 1] != b[:, 1]).sum()))
