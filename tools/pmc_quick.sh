#!/bin/bash
# quick PMC pass on the bench workload: instruction-class counters of the solve kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-pmcq}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/${TAG}_a -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_a.err; echo "a exit=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/${TAG}_b -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_b.err; echo "b exit=$?"
python3 - <<PY
import csv, collections, glob
for sub in ("${TAG}_a", "${TAG}_b"):
    for f in glob.glob("$OUT/%s/pmc_counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mpc_solve" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(sub, k, "%.4g" % (sum(v) / len(v)))
PY
