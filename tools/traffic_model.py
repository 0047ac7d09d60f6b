#!/usr/bin/env python3
"""Bytes the ACTIVE lanes of the headline launch ask for, against what the counters say was moved.

The test-only host build of the solver header (tests/host_twin) runs the bench's 65 536 instances on a workspace that counts
every record the device's sweeps fetch (LDS-DMA) or store; instances are grouped 64 to a wave in launch order and a wave's
pass p touches the lanes that still run.  Printed: the bytes per solve, the sum over the launch, and what the launch would
fetch if memory were read in sectors of 32 / 64 / 128 B (2 / 4 / 8 neighbouring lanes of a 16-byte group) or whole rows --
to be read beside FETCH_SIZE / WRITE_SIZE with and without lane compaction (profiles/r03_lane_compact.json).  CPU only (about a minute); writes JSON to stdout."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pkg = G.load_package()
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "host_twin")])
    tw = C.CDLL(os.path.join(ROOT, "tests", "host_twin", "libhost_twin.so"))
    gd = os.path.join(ROOT, "tests", "golden")
    wp = pkg.scenarios.load_waypoints(os.path.join(gd, "lake_track_waypoints.csv"))
    p = pkg.params_from_json(os.path.join(gd, "config-fast.json"))
    b = pkg.scenarios.lake_track_batch(B, p, wp, seed=1234)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    st, cf, yl, yh = f(b["state"]), f(b["coeffs"]), f(b["yaw_lo"]), f(b["yaw_hi"])
    counts = np.zeros((B, 4), dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = tw.mpc_host_twin_traffic(C.byref(p), C.c_int64(B), C.c_int64(B), vp(st), vp(cf), vp(yl), vp(yh), None, vp(counts))
    assert rc == 0
    fetched, stored, passes, iters = (counts[:, q].astype(np.float64) for q in range(4))
    io_in, io_out = 8 * 13, 8 * (9 + 2 * p.N) + 8
    out = {
        "batch": B, "mean_iterations": iters.mean(), "mean_passes": passes.mean(),
        "workspace_bytes_fetched_per_solve": 8 * fetched.mean(), "workspace_bytes_stored_per_solve": 8 * stored.mean(),
        "bytes_fetched_per_pass": 8 * fetched.sum() / passes.sum(), "bytes_stored_per_pass": 8 * stored.sum() / passes.sum(),
        "launch_fetch_GB_active_lanes": (8 * fetched.sum() + io_in * B) / 1e9, "launch_write_GB_active_lanes": (8 * stored.sum() + io_out * B) / 1e9,
    }
    # a wave's pass touches the sectors that hold a running lane; fetch per lane and pass taken as the instance's own mean
    W = B // 64
    per_pass = (8 * fetched / np.maximum(passes, 1)).reshape(W, 64)
    P = passes.reshape(W, 64).astype(np.int64)
    wave_passes = int(P.max(1).sum())
    out["wave_passes"] = wave_passes
    out["lanes_active_mean"] = float(P.sum() / wave_passes)
    mean_pp = 8 * fetched.sum() / passes.sum()
    for lanes in (2, 4, 8, 64):
        tot = 0.0
        for w in range(W):
            row = P[w]
            for ps in range(int(row.max())):
                act = row > ps
                tot += act.reshape(-1, lanes).any(1).sum() * lanes * mean_pp
        out["launch_fetch_GB_if_%dB_sectors" % (16 * lanes)] = (tot + io_in * B) / 1e9
    # the same lanes packed perfectly into the fewest lines at every pass: what any compaction scheme could reach at best
    tot = 0.0
    for w in range(W):
        row = P[w]
        for ps in range(int(row.max())):
            tot += -(-int((row > ps).sum()) // 8) * 8 * mean_pp
    out["launch_fetch_GB_if_128B_lines_perfectly_packed"] = (tot + io_in * B) / 1e9
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r03_lane_compact.json")))["counters_per_launch_headline_fp64_65536"]
        out["measured_fetch_GB_lane_compact_0_and_2"] = pm["fetched_GB"]
        out["measured_write_GB_lane_compact_0_and_2"] = pm["written_GB"]
        out["measured_over_active_lanes_fetch"] = [v / out["launch_fetch_GB_active_lanes"] * (B / 65536) for v in pm["fetched_GB"]]
    except Exception as e:  # noqa: BLE001
        out["measured"] = "profiles/r03_lane_compact.json not readable: %s" % e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
