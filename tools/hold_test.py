#!/usr/bin/env python3
"""GPU-box helper: duration of the bulk launches on the survey population with the tail slices held back (MPC_TAIL_HOLD_BATCHES) and
with them running beside the launches -- what the slices cost the launches."""
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
pipe = bench.Pipeline(pkg, torch, params, B, tens, None, True, args.inflight, dev, 0, None, args, tail_cut=args.tail_cut, tail_ring=args.tail_ring, outstanding=256)
hold = int(os.environ.get("MPC_TAIL_HOLD_BATCHES", "0"))
n = 2 * hold if hold else 40
t0 = time.perf_counter()
pipe.run_until(lambda p: p.n_issued >= n)
torch.cuda.synchronize()
el = time.perf_counter() - t0
k = pipe.kernel_ms(4, n)
print("population %s hold %d: %d launches issued in %.1f ms (%.3f ms per batch, bulk only); launch duration mean %.3f ms median %.3f" % (args.population, hold, n, 1e3 * el, 1e3 * el / n, np.mean(k), np.median(k)), flush=True)
pipe.drain(); torch.cuda.synchronize(); pipe.close()
