#!/bin/bash
# stall breakdown of the solve kernel at a given batch:  bash tools/pmc_stall.sh <tag> <batch>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-pmcs}; B=${2:-65536}
cd /tmp && export TMPDIR=/tmp
run() { local n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > /dev/null 2> $OUT/${TAG}_$n.err; echo "$n exit=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU && \
run b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python3 - <<PY
import csv, collections, glob
for sub in ("a", "b"):
    for f in glob.glob("$OUT/${TAG}_%s/pmc_counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mpc_solve" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(sub, k, "%.5g" % (sum(v) / len(v)))
PY
