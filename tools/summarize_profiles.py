#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/gpu_round.sh (gpurun_out/<tag>_*) into profiles/<name>_*.
   python tools/summarize_profiles.py <tag> <name>"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

shutil.copy(os.path.join(G, tag + "_prof", "trace_kernel_stats.csv"), os.path.join(P, name + "_kernel_stats.csv"))
bench = json.load(open(os.path.join(G, tag + "_bench.json")))
json.dump(bench, open(os.path.join(P, name + "_bench.json"), "w"), indent=1)


GRID = (bench["config"]["batch_per_gpu"] + 63) // 64 * 64


def short(kname):
    """mpc_solve_kernel<true, double, 1, double, float> -> 'solve<double,1,double,float>' (solver reals, waves/SIMD, ABI reals, source reals)"""
    i = kname.find("<")
    args = kname[i + 1:kname.rfind(">")].replace(" ", "").split(",") if i >= 0 else []
    base = "tail" if "mpc_tail" in kname else "solve"
    return base + "<" + ",".join(args[1:]) + ">"


def pmc(sub, kernel):
    rows = list(csv.DictReader(open(os.path.join(G, tag + "_" + sub, "pmc_counter_collection.csv"))))
    agg = collections.defaultdict(list)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in rows:
        # only the launches of the benchmarked batch (the bench also makes small launches, e.g. its B = 1 latency leg)
        if kernel in r["Kernel_Name"] and int(r["Grid_Size"]) == GRID:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    phases = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items()}
    if len(phases) > 1:       # a mixed-precision solve is two launches per batch: per batch = the sum over its phases
        tot = collections.defaultdict(float)
        for d in phases.values():
            for c, v in d.items():
                tot[c] += v
        return dict(tot), {k: len(v) for k, v in agg.items()}, meta, phases
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}, meta, phases


out = {"source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (separate passes)",
       "kernel": "mpc_solve_kernel", "per": "launch (mean over the profiled launches)"}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    m, n, meta, phases = pmc(sub, "mpc_solve_kernel")
    out.update(m); out.setdefault("launches", {}).update(n)
    if len(phases) > 1:
        out["per"] = "batch = the sum over the phases of a mixed-precision solve (each phase: mean over its profiled launches)"
        for k, d in phases.items():
            out.setdefault("phases", {}).setdefault(k, {}).update(d)
    # rocprofv3's per-dispatch fields, as reported: VGPR_Count is the arch-VGPR allocation granule-rounded and Accum/LDS are
    # not filled in for this kernel on ROCm 7.2 -- the kernel's real footprint is in tools/isa_report.py (ISA metadata)
    out["dispatch_fields_as_reported_by_rocprofv3"] = meta
# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count with the same access width (tools/calib_fetch.hip)
cal = {}
for sub, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    f = os.path.join(G, tag + "_" + sub, "pmc_counter_collection.csv")
    if os.path.exists(f):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "calib_copy8" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
        if vals:
            cal[ctr] = {"counter_kb_per_launch": sum(vals) / len(vals), "true_bytes_per_launch": (1 << 28) * 8,
                        "bytes_per_counter_kb": (1 << 28) * 8 / (sum(vals) / len(vals))}
out["calibration_8B_per_lane"] = cal
fk = cal.get("FETCH_SIZE", {}).get("bytes_per_counter_kb", 1024.0)
wk = cal.get("WRITE_SIZE", {}).get("bytes_per_counter_kb", 1024.0)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_bytes_per_launch"] = out["FETCH_SIZE"] * fk + out["WRITE_SIZE"] * wk
    out["hbm_bytes_note"] = "FETCH_SIZE and WRITE_SIZE are in KB; scaled by the calibration above when present (else x1024)"
if "SQ_WAVE_CYCLES" in out:
    out["wait_any_fraction_of_wave_cycles"] = out.get("SQ_WAIT_ANY", 0) / out["SQ_WAVE_CYCLES"]
    out["active_inst_fraction_of_wave_cycles"] = out.get("SQ_ACTIVE_INST_ANY", 0) / out["SQ_WAVE_CYCLES"]
json.dump(out, open(os.path.join(P, name + "_pmc_summary.json"), "w"), indent=1)
cfg = bench["config"]
entry = {"batch": cfg["batch_per_gpu"], "config": "config-fast.json" if "config-fast" in cfg["workload"] else "config-stable.json", "N": cfg["N"],
         "dtype": bench["dtype"], "weights_sweep": "weight sweep" in cfg["workload"], "traj": "trajectories on" in cfg["workload"],
         "hbm_bytes_per_launch": out.get("hbm_bytes_per_launch"), "from": name + "_pmc_summary.json",
         "mixed_precision": cfg.get("mixed_precision", "no")}
tf = os.path.join(P, name.split("_")[0] + "_pmc_traffic.json")
allt = json.load(open(tf)) if os.path.exists(tf) else {"entries": []}
allt["entries"] = [e for e in allt["entries"] if e["from"] != entry["from"]] + [entry]
json.dump(allt, open(tf, "w"), indent=1)
print(json.dumps(out, indent=1))
