#!/bin/bash
# PMC passes that separate instruction-fetch stalls from memory and LDS waits of the solve kernel.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-pmci}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/${TAG}_avail.txt 2>&1
run() { # name counters...
  local n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_$n.err; echo "$n exit=$?"
}
run a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES && \
run b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU && \
run c SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM
python3 - <<PY
import csv, collections, glob
for sub in ("a", "b", "c"):
    for f in glob.glob("$OUT/${TAG}_%s/pmc_counter_collection.csv" % sub):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mpc_solve" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(sub, k, "%.5g" % (sum(v) / len(v)))
PY
