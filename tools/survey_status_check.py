#!/usr/bin/env python3
"""GPU-box tool: SURVEY.md 8d's populations (configs[2] 65 536 at N = 10; configs[3]'s share 32 768 at N = 25; the 65 536-instance fp64
weight sweep) solved on the DEVICE twice -- in single launches and with MPC_TAIL_AUTO (tail slices) -- and by the ORACLE on every
hard instance (more than 22 iterations or not converged) plus a random sample, the oracle spread over the host's cores.
Writes one JSON line per population: status counts, device-vs-device bitwise equality, device-vs-oracle status agreement, longest
chain.   python tools/survey_status_check.py > gpurun_out/r04_survey_status.jsonl"""
import json, os, sys, time
import multiprocessing as mp
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def oracle_job(job):
    import oracle_lib as O
    name, over, st, cf, yl, yh, w, max_iter = job
    cfg = O.load_config(name, **over)
    opt = O.default_options(max_iter=max_iter)
    res = []
    for i in range(st.shape[1]):
        cfg.yaw_low, cfg.yaw_high = float(yl[i]), float(yh[i])
        if w is not None:
            for q in range(12):
                cfg.weights[q] = float(w[q, i])
        s, o9, tx, ty, info = O.mpc_solve(cfg, st[:, i], cf[:, i], opt)
        res.append((s, info.iterations, info.acceptable_restored_older, info.no_restart, list(o9)))
    return res


def main():
    import torch
    import __graft_entry__ as G
    pkg = G.load_package()
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    golden = os.path.join(ROOT, "tests", "golden")
    wp = pkg.scenarios.load_waypoints(os.path.join(golden, "lake_track_waypoints.csv"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    cores = max(1, min(60, len(os.sched_getaffinity(0)) - 2))
    for name, config, over, B, sweep in (("configs[2], N=10", "config-fast.json", {}, 65536, False),
                                         ("configs[3] share, N=25", "config-stable.json", dict(N=25, dt=0.05), 32768, False),
                                         ("fp64 weight sweep, N=10", "config-fast.json", {}, 65536, True)):
        params = pkg.params_from_json(os.path.join(golden, config), **over)
        b = pkg.scenarios.lake_track_batch(B, params, wp, stream=3, filtered="survey")
        w = pkg.scenarios.weight_sweep(B, params, seed=1234, velocity_weights=(0.0, 1.0, 100.0)) if sweep else None
        ins = (t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]))
        d_w = t(w) if w is not None else None
        res = {}
        for mode, cut in (("single launch", 0), ("tail slices", -1)):
            p = params.copy(); p.tail_cut = cut
            with pkg.BatchedMPC(p, B, device=0) as mpc:
                t0 = time.perf_counter()
                r = mpc.solve_torch(*ins, weights=d_w, want_traj=True)
                mpc.tail_wait(0); torch.cuda.synchronize()
                el = time.perf_counter() - t0
                res[mode] = {k: v.cpu().numpy() for k, v in r.items()}
                res[mode]["seconds"] = el
                res[mode]["info"] = mpc.tail_info() if cut else None
        a, c = res["single launch"], res["tail slices"]
        bitwise = all(np.array_equal(a[k], c[k], equal_nan=True) for k in ("status", "iters", "out", "traj"))
        it, st = a["iters"], a["status"]
        hard = np.where((it > 22) | (st != 0))[0]
        idx = np.unique(np.concatenate([hard, np.random.default_rng(1).choice(B, 512, replace=False)]))
        chunks = np.array_split(idx, cores * 4)
        jobs = [(config, over, b["state"][:, ch].copy(), b["coeffs"][:, ch].copy(), b["yaw_lo"][ch].copy(), b["yaw_hi"][ch].copy(), None if w is None else w[:, ch].copy(), params.max_iter)
                for ch in chunks if len(ch)]
        t1 = time.perf_counter()
        with mp.get_context("spawn").Pool(cores) as pool:
            flat = [x for ch in pool.map(oracle_job, jobs) for x in ch]
        ost = np.array([x[0] for x in flat]); oit = np.array([x[1] for x in flat]); oo = np.array([x[4] for x in flat]).T
        mism = np.where(ost != st[idx])[0]
        same = (ost == st[idx]) & (ost == 0)
        d = np.abs(oo[:8] - a["out"][:8, idx])
        close = same & (np.abs(oo[6] - a["out"][6, idx]) <= 1e-6)
        print(json.dumps({"population": name + ", SURVEY 8d's rejection only", "instances": B,
                          "status_device": np.bincount(st, minlength=7).tolist(), "longest_chain_iterations": int(it.max()),
                          "instances_above_40_80_150_iterations": [int((it > 40).sum()), int((it > 80).sum()), int((it > 150).sum())],
                          "tail_slices_bitwise_equal_to_single_launch": bool(bitwise),
                          "single_launch_seconds": a["seconds"], "tail_slices_seconds_one_batch_to_final": c["seconds"], "tail_info": c["info"],
                          "oracle_instances": int(len(idx)), "oracle_hard_instances": int(len(hard)), "oracle_seconds": time.perf_counter() - t1, "oracle_workers": cores,
                          "status_oracle_on_those": np.bincount(ost, minlength=7).tolist(),
                          "status_differences_device_vs_oracle": [(int(idx[j]), int(st[idx[j]]), int(ost[j]), int(it[idx[j]]), int(oit[j])) for j in mism],
                          "oracle_restored_an_older_acceptable_point": int(sum(x[2] for x in flat)), "oracle_no_restart_cases": int(sum(x[3] for x in flat)),
                          "converged_in_both": int(same.sum()), "of_those_steer_within_1e-6": int(close.sum()),
                          "max_abs_dsteer_where_close": float(np.abs(oo[6] - a["out"][6, idx])[close].max()) if close.any() else None,
                          "forks_to_another_local_minimum": int(same.sum() - close.sum())}), flush=True)


if __name__ == "__main__":
    main()
