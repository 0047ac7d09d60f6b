#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line (GPU-box helper)."""
import json, sys
for p in sys.argv[1:]:
    try:
        r = json.load(open(p))
    except Exception as e:
        print(p, "unreadable:", e); continue
    rf = r.get("roofline", {})
    print("  %s: %.4g solves/s  %.3f ms/step  dtype %s  B %d  inflight %s  iters %.2f (max %d)  conv %.5f  status %s" % (
        p.split("/")[-1], r["value"], r["ms_per_step"], r["dtype"], r["config"]["batch_per_gpu"], r["config"].get("batches_in_flight"),
        r["mean_iterations"], r["max_iterations"], r["converged_fraction"], r["status_counts"]))
    print("      kernel avg %.3f alone %s ms  valu_frac %.3f  hbm frac %.2e  d_steer %s d_acc %s" % (
        rf.get("kernel_ms_avg", 0), rf.get("kernel_ms_alone"), rf.get("valu_frac", 0), rf.get("frac", 0), r.get("max_abs_dsteer_vs_oracle"), r.get("max_abs_daccel_vs_oracle")))
    if "host_path" in r:
        h = r["host_path"]; print("      host path: %.4g solves/s incl PCIe (%.2f ms/batch, bitwise %s); B=1 latency %.3f ms (kernel %.3f ms, %d iters)" % (
            h["solves_per_s_incl_pcie"], h["ms_per_batch"], h["matches_device_path_bitwise"], h["b1_latency_ms_median"], h["b1_kernel_ms"], h["b1_iterations"]))
    if "cpu_baseline" in r:
        print("      cpu: %.1f solves/s on 1 core; %.1f on %d cores" % (r["cpu_baseline"]["value"], r["cpu_baseline_all_cores"]["value"], r["cpu_baseline_all_cores"]["cores"]))
    for k in ["unfiltered"] + sorted(r.get("other_configs", {})):
        l = r.get(k) or r.get("other_configs", {}).get(k)
        if l:
            print("      leg %-16s %.4g solves/s  %.3f ms/batch  B %d  inflight %d  tail_cut %s  iters %.2f (max %d)  status %s" % (
                k, l["solves_per_s"], l["ms_per_batch"], l["batch"], l["batches_in_flight"], l["tail_cut"], l["mean_iterations"], l["max_iterations"],
                {a: b for a, b in l["status_counts"].items() if b}))
