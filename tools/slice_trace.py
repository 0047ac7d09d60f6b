#!/usr/bin/env python3
"""Reads an MPC_TAIL_TRACE file (one line per retired tail slice, written by libmpc_amd.so) and prints what the slices did: their
cadence per handle, how full they were, what share of the entries they touched finished / moved on untouched / was parked again.
   python tools/slice_trace.py gpurun_out/trace.txt [skip_first_n_slices_per_handle]"""
import sys, collections
import numpy as np
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    f = l.split()
    if len(f) != 15: continue
    rows[f[0]].append([int(x) for x in f[1:]])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for h, r in rows.items():
    a = np.array(r)[skip:]
    if len(a) < 3: continue
    k, t0, t1, waves, nabs, surv_in, fresh, surv_out, fin, moved, reparked, maxwp, notfinal, inflight = a.T
    span = (t1[-1] - t1[0]) / 1e3
    cad = np.diff(t1) / 1e3
    print("handle %s: %d slices over %.1f ms: a slice retired every %.2f ms (median %.2f, p90 %.2f); issue->retire %.2f ms median" %
          (h[-6:], len(a), span, span / (len(a) - 1), np.median(cad), np.percentile(cad, 90), np.median(t1 - t0) / 1e3))
    print("   waves mean %.0f (max %d); fresh queues per slice %.2f; fresh entries per slice %.0f; survivors in (known at issue) %.0f, out %.0f"
          % (waves.mean(), waves.max(), nabs.mean(), fresh.mean(), surv_in.mean(), surv_out.mean()))
    tot = fin + moved + reparked
    print("   per slice: finished %.0f, moved on untouched %.0f, parked again %.0f; most passes of a wave: mean %.1f max %d; batches not final: mean %.0f max %d"
          % (fin.mean(), moved.mean(), reparked.mean(), maxwp.mean(), maxwp.max(), notfinal.mean(), notfinal.max()))
    print("   lane use: entries handled per lane %.2f; survivors out per wave lane %.2f" % ((fin + reparked).sum() / (64.0 * waves.sum()), surv_out.sum() / (64.0 * waves.sum())))
