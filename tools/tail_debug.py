#!/usr/bin/env python3
"""GPU-box helper: the survey population through bench.Pipeline for a bounded time, with a progress line per 0.5 s
(batches issued / final, and each handle's tail counters)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench, __graft_entry__ as G
pkg = G.load_package()
args = bench.parse_args(sys.argv[1:])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
golden = os.path.join(ROOT, "tests", "golden")
params = pkg.params_from_json(os.path.join(golden, args.config))
wp = pkg.scenarios.load_waypoints(os.path.join(golden, "lake_track_waypoints.csv"))
B = args.batch
b = pkg.scenarios.lake_track_batch(B, params, wp, stream=3, filtered={"filtered": True, "survey": "survey", "unfiltered": False}[args.population])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tens = (t(b["state"]), t(b["coeffs"]), t(b["yaw_lo"]), t(b["yaw_hi"]))
pipe = bench.Pipeline(pkg, torch, params, B, tens, None, True, args.inflight, dev, 0, None, args, tail_cut=args.tail_cut, tail_ring=args.tail_ring,
                      outstanding=min(256, args.inflight * args.tail_ring))
t0 = time.perf_counter(); last = [t0]
limit = float(os.environ.get("DEBUG_SECONDS", "12"))
def cond(p):
    now = time.perf_counter()
    if now - last[0] > 0.5:
        last[0] = now
        info = [h.tail_info() for h in p.mpcs]
        print("t=%5.1f issued %5d final %5d launches %d | %s" % (now - t0, p.n_issued, p.final_upto, len(p.launches),
              " | ".join("slices %d surv %d cut %s share %s thr %d" % (i["tail_launches"], i["survivors"], i["tail_cut_in_use"], i["deferred_share"], i["batches_not_deferred_survivors_full"]) for i in info)), flush=True)
    return now - t0 > limit
pipe.run_until(cond)
n, tt = pipe.final_upto, time.perf_counter() - t0
print("draining...", flush=True)
pipe.drain(); torch.cuda.synchronize()
print("done: %d batches final in %.2f s = %.2f M solves/s; drained at %.2f s" % (n, tt, n * B / tt / 1e6, time.perf_counter() - t0))
pipe.close()
