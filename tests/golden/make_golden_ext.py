#!/usr/bin/env python3
"""Generate tests/golden/scipy_cross_solve_ext.json: the independent SLSQP cross-solve of make_golden.py for the
other two shapes of BASELINE.json -- the long horizon (configs[3]: N=25, dt=0.05) and per-instance weights
(configs[4]: the values swept in submission-report.md:303-319).  Same independent numpy restatement of the NLP,
same solver; see make_golden.py.

Run:  python tests/golden/make_golden_ext.py     (a few minutes)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as MG


def main():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    pkg = G.load_package()   # scenario generators (pure numpy) and the JSON loader only
    wp = pkg.scenarios.load_waypoints(os.path.join(HERE, "lake_track_waypoints.csv"))
    cases = []
    # long horizon
    p25 = pkg.params_from_json(os.path.join(HERE, "config-fast.json"), N=25, dt=0.05)
    lb = pkg.scenarios.lake_track_batch(6, p25, wp, seed=79)
    for i in range(6):
        cases.append(dict(name="lake-N25-%d" % i, config="config-fast.json", N=25, dt=0.05, weights=None,
                          state=lb["state"][:, i].tolist(), coef=lb["coeffs"][:, i].tolist(),
                          yaw_lo=float(lb["yaw_lo"][i]), yaw_hi=float(lb["yaw_hi"][i])))
    # per-instance weights
    p10 = pkg.params_from_json(os.path.join(HERE, "config-fast.json"))
    lw = pkg.scenarios.lake_track_batch(10, p10, wp, seed=80)
    ww = pkg.scenarios.weight_sweep(10, p10, seed=80)
    for i in range(10):
        cases.append(dict(name="lake-weights-%d" % i, config="config-fast.json", N=10, dt=0.1, weights=ww[:, i].tolist(),
                          state=lw["state"][:, i].tolist(), coef=lw["coeffs"][:, i].tolist(),
                          yaw_lo=float(lw["yaw_lo"][i]), yaw_hi=float(lw["yaw_hi"][i])))
    out = []
    for cs in cases:
        c = MG.load_config(cs["config"])
        c["N"], c["dt"] = cs["N"], cs["dt"]
        if cs["weights"] is not None:
            c["w"] = list(cs["weights"])
        P = MG.Problem(c, cs["state"], cs["coef"], cs["yaw_lo"], cs["yaw_hi"])
        o9, z, viol, ok = P.solve()
        print("%-18s ok=%s viol=%.1e delta0=%.9f a0=%.6f J=%.6f" % (cs["name"], ok, viol, o9[6], o9[7], o9[8]), flush=True)
        cs = dict(cs); cs["out9"] = o9.tolist(); cs["constraint_violation"] = viol; cs["slsqp_success"] = ok
        cs["cost_at_xi"] = float(P.f(P.xi))
        out.append(cs)
        json.dump({"generator": "tests/golden/make_golden_ext.py", "solver": "scipy SLSQP + complex-step derivatives",
                   "cases": out}, open(os.path.join(HERE, "scipy_cross_solve_ext.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
