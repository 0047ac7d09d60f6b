#!/usr/bin/env python3
"""Generate tests/golden/scipy_cross_solve.json: an INDEPENDENT cross-solve of the reference NLP.

The reference (dr-tony-lin/CarND-MPC-Project, src/control/MPC.cpp) cannot be built or run in this
image (CppAD / IPOPT / MUMPS absent, no network) and its repository holds no numeric golden vectors
for MPC::solve().  This script therefore restates the NLP a second time, in numpy, independently of
oracle/mpc_oracle.c and of the HIP code, and solves it with scipy's SLSQP (a different algorithm
family from both: active-set SQP, not interior point).  Jacobians come from complex-step
differentiation of the residual function, so they do not share hand-derived formulas with the C
code either.  The stored vectors pin:  inputs -> (x1,y1,psi1,v1,cte1,epsi1,delta0,a0,J*).

NLP (frozen-tape semantics): variables in the reference's quantity-major layout
(MPC.cpp:56-63), objective MPC.cpp:68-114 with every `if` decided at the start point xi, residuals
MPC.cpp:116-153, bounds MPC.cpp:222-257, start point MPC.cpp:207-218.

Run:  python tests/golden/make_golden.py     (needs scipy; takes about a minute)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def mph2mps(x):
    return x * 1609.34 / 3600.0


def load_config(name):
    """Config::load, src/utils/Config.cpp:31-87 (only what the solve path reads)."""
    js = json.load(open(os.path.join(HERE, name)))
    c = {"N": int(js["N"]), "dt": js["dt"], "Lf": js["Lf"], "w": list(js["weights"]),
         "cte_panic": js["cte panic"], "epsi_panic": js["epsi panic"],
         "max_steering": js["max steering"] * np.pi / 180, "max_acc": mph2mps(js["max acceleration"]),
         "max_dec": mph2mps(js["max deceleration"]), "max_speed": mph2mps(js["max speed"]),
         "steers": list(js["steers"])}
    scale = c["max_speed"] / mph2mps(100.0)
    c["steer_speeds"] = [min(mph2mps(s), c["max_speed"]) if scale <= 1 else mph2mps(s) * scale
                         for s in js["steer speeds"]]
    return c


def speed_target(c, angle, maxv):
    y = abs(angle)
    for i, s in enumerate(c["steers"]):
        if y <= s:
            return min(c["steer_speeds"][min(i, len(c["steer_speeds"]) - 1)], maxv)
    return min(c["steer_speeds"][-1], maxv)


class Problem:
    def __init__(self, c, state, coef, yaw_lo, yaw_hi):
        self.c, self.state, self.coef = c, np.asarray(state, float), np.asarray(coef, float)
        N = self.N = c["N"]
        self.ix = dict(x=0, y=N, psi=2 * N, v=3 * N, cte=4 * N, epsi=5 * N, delta=6 * N, a=7 * N - 1)
        self.n = 8 * N - 2
        xi = np.zeros(self.n)
        for k, q in enumerate(("x", "y", "psi", "v", "cte", "epsi")):
            xi[self.ix[q]] = self.state[k]
        self.xi = xi
        I = self.ix
        w = c["w"]
        # branch outcomes at xi (CppAD records the tape once, at xi)
        self.wcte = np.where(np.abs(xi[I["cte"]:I["cte"] + N]) < c["cte_panic"], w[0], w[11])
        self.wepsi = np.where(np.abs(xi[I["epsi"]:I["epsi"] + N]) > c["epsi_panic"], w[10], w[1])
        self.vref = np.array([speed_target(c, xi[I["psi"] + i], c["max_speed"]) for i in range(N)])
        self.negv = xi[I["v"]:I["v"] + N] < 0
        a0 = xi[I["a"]:I["a"] + N - 1]
        self.apos = a0 > 0
        self.declow = (a0 < 0) & (xi[I["v"]:I["v"] + N - 1] < a0)
        self.dapos = a0[1:] > a0[:-1]
        lo = np.full(self.n, -np.inf)
        hi = np.full(self.n, np.inf)
        lo[I["psi"]:I["psi"] + N] = yaw_lo; hi[I["psi"]:I["psi"] + N] = yaw_hi
        lo[I["v"]:I["v"] + N] = -c["max_speed"]; hi[I["v"]:I["v"] + N] = c["max_speed"]
        lo[I["delta"]:I["delta"] + N - 1] = -c["max_steering"]; hi[I["delta"]:I["delta"] + N - 1] = c["max_steering"]
        lo[I["a"]:I["a"] + N - 1] = c["max_dec"]; hi[I["a"]:I["a"] + N - 1] = c["max_acc"]
        self.lo, self.hi = lo, hi

    def split(self, z):
        I, N = self.ix, self.N
        return (z[I["x"]:I["x"] + N], z[I["y"]:I["y"] + N], z[I["psi"]:I["psi"] + N], z[I["v"]:I["v"] + N],
                z[I["cte"]:I["cte"] + N], z[I["epsi"]:I["epsi"] + N], z[I["delta"]:I["delta"] + N - 1],
                z[I["a"]:I["a"] + N - 1])

    def f(self, z):
        x, y, psi, v, cte, epsi, d, a = self.split(z)
        w = self.c["w"]
        f = np.sum(self.wcte * cte ** 2 + self.wepsi * epsi ** 2 + w[2] * (v - self.vref) ** 2)
        f += np.sum(np.where(self.negv, w[9] * v ** 2, 0.0))
        f += np.sum(w[3] * d ** 2) + np.sum(np.where(self.apos, w[6] * a ** 2, 0.0))
        f += np.sum(np.where(self.declow, w[8] * (v[:-1] - a) ** 2, 0.0))
        f += np.sum(w[4] * (d[1:] - d[:-1]) ** 2) + np.sum(np.where(self.dapos, w[7] * (a[1:] - a[:-1]) ** 2, 0.0))
        return f

    def grad(self, z):
        # complex-step gradient (exact to rounding; the objective is a polynomial in z)
        g = np.zeros(self.n)
        h = 1e-30
        for i in range(self.n):
            zz = z.astype(complex); zz[i] += 1j * h
            g[i] = self.f(zz).imag / h
        return g

    def g(self, z):
        x, y, psi, v, cte, epsi, d, a = self.split(z)
        dt, Lf = self.c["dt"], self.c["Lf"]
        co = self.coef
        fx = sum(co[k] * x[:-1] ** k for k in range(len(co)))
        fpx = sum(k * co[k] * x[:-1] ** (k - 1) for k in range(1, len(co)))
        vdt = v[:-1] * dt
        psin = psi[:-1] + d * vdt / Lf
        r = [z[[self.ix[q] for q in ("x", "y", "psi", "v", "cte", "epsi")]] - self.state,
             x[1:] - (x[:-1] + np.cos(psi[:-1]) * vdt),
             y[1:] - (y[:-1] + np.sin(psi[:-1]) * vdt),
             psi[1:] - psin,
             v[1:] - (v[:-1] + a * dt),
             cte[1:] - ((fx - y[:-1]) + np.sin(epsi[:-1]) * vdt),
             epsi[1:] - (psin - np.arctan(fpx))]
        return np.concatenate(r)

    def jac(self, z):
        h = 1e-30
        m = len(self.g(z))
        J = np.zeros((m, self.n))
        for i in range(self.n):
            zz = z.astype(complex); zz[i] += 1j * h
            J[:, i] = self.g(zz).imag / h
        return J

    def solve(self):
        bounds = [(None if not np.isfinite(l) else l, None if not np.isfinite(u) else u) for l, u in zip(self.lo, self.hi)]
        cons = [{"type": "eq", "fun": self.g, "jac": self.jac}]
        z = self.xi
        ok = False
        # SLSQP stops on the relative change of the objective: restart it from its own solution
        # (first on a down-scaled objective, which converges fastest from the cold start, then
        # unscaled) until the point stops moving.  Its usual final message is "positive
        # directional derivative for linesearch" = no further progress possible, which is fine.
        import warnings
        for rep, k in enumerate((1e-2, 1.0, 1.0, 1.0)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = minimize(lambda q: self.f(q) * k, z, jac=lambda q: self.grad(q) * k, method="SLSQP",
                               bounds=bounds, constraints=cons, options={"maxiter": 400, "ftol": 1e-15})
            moved = float(np.max(np.abs(res.x - z)))
            z = res.x
            ok = bool(res.success) or res.status == 8
            if rep > 0 and moved < 1e-9:
                break
        I = self.ix
        out9 = [z[I["x"] + 1], z[I["y"] + 1], z[I["psi"] + 1], z[I["v"] + 1], z[I["cte"] + 1], z[I["epsi"] + 1],
                z[I["delta"]], z[I["a"]], self.f(z)]
        return np.array(out9), z, float(np.max(np.abs(self.g(z)))), ok


def main():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    pkg = G.load_package()   # only for the scenario generators (pure numpy) and the JSON loader
    cases = []
    # (1) the reference's own demo scenario, src/test.cpp:45-50, inputs as recorded in BASELINE.md section 2
    c = load_config("config-stable.json")
    cases.append(dict(name="test.cpp:45-50", config="config-stable.json",
                      state=[0.0, 0.0, 0.0, 26.6806, -0.1765561004933686, 0.02598647310119508],
                      coef=[-0.1765561004933686, -0.02599232420893351, 0.003029165833778106, 0.0, 0.0],
                      yaw_lo=-0.1, yaw_hi=0.48434539231718154))
    # (2) lake-track instances, config-fast.json (BASELINE.json configs[2])
    pf = pkg.params_from_json(os.path.join(HERE, "config-fast.json"))
    wp = pkg.scenarios.load_waypoints(os.path.join(HERE, "lake_track_waypoints.csv"))
    lb = pkg.scenarios.lake_track_batch(24, pf, wp, seed=77)
    for i in range(24):
        cases.append(dict(name="lake-%d" % i, config="config-fast.json", state=lb["state"][:, i].tolist(),
                          coef=lb["coeffs"][:, i].tolist(), yaw_lo=float(lb["yaw_lo"][i]), yaw_hi=float(lb["yaw_hi"][i])))
    # (3) straight-line offsets, config-stable.json (BASELINE.json configs[1])
    ps = pkg.params_from_json(os.path.join(HERE, "config-stable.json"))
    sb = pkg.scenarios.straight_line_batch(12, ps, seed=78)
    for i in range(12):
        cases.append(dict(name="straight-%d" % i, config="config-stable.json", state=sb["state"][:, i].tolist(),
                          coef=sb["coeffs"][:, i].tolist(), yaw_lo=-0.1, yaw_hi=0.1))
    out = []
    for cs in cases:
        c = load_config(cs["config"])
        P = Problem(c, cs["state"], cs["coef"], cs["yaw_lo"], cs["yaw_hi"])
        o9, z, viol, ok = P.solve()
        print("%-16s ok=%s viol=%.1e delta0=%.9f a0=%.6f J=%.6f" % (cs["name"], ok, viol, o9[6], o9[7], o9[8]), flush=True)
        cs = dict(cs); cs["out9"] = o9.tolist(); cs["constraint_violation"] = viol; cs["slsqp_success"] = ok
        cs["cost_at_xi"] = float(P.f(P.xi))
        out.append(cs)
    json.dump({"generator": "tests/golden/make_golden.py", "solver": "scipy SLSQP + complex-step derivatives",
               "cases": out}, open(os.path.join(HERE, "scipy_cross_solve.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
