#!/usr/bin/env python3
"""Mean of every counter over the mpc_solve_kernel dispatches of the rocprofv3 --pmc output directories given on the command line
(grid of at least 1024 workgroups x 64 by default: the full-batch launches).  Prints one JSON object per directory."""
import csv
import glob
import json
import os
import sys

for d in sys.argv[1:]:
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mpc_solve_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 16384:
                continue
            kind = "f32" if "mpc_solve_kernel<true, float" in r["Kernel_Name"] else "f64"
            acc.setdefault((kind, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    out = {"dir": os.path.basename(d.rstrip("/"))}
    for (kind, name), v in sorted(acc.items()):
        out["%s %s" % (kind, name)] = sum(v) / len(v)
        out["%s launches" % kind] = len(v)
    print(json.dumps(out))
