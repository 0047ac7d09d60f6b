#!/bin/bash
# what does a lone wave wait for?  stall counters of the solve kernel at B = 64 (one wave) -- the serial chain of B = 1 and of every straggler
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out
run() { local n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/r02r_$n -o pmc -- python3 $R/bench.py --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --no-host-leg --batch 64 > /dev/null 2> $OUT/r02r_$n.err; echo "$n exit=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU && \
run b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU && \
run c SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
python3 - <<PY
import csv, collections, glob
for sub in ("a", "b", "c"):
    for f in glob.glob("$OUT/r02r_%s/**/pmc_counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mpc_solve" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(sub, k, "%.5g" % (sum(v) / len(v)), len(v))
PY
python3 bench.py --steps 50 --warmup 5 --inflight 1 --no-cpu-baseline --no-host-leg --batch 64 > gpurun_out/r02r_b64.json 2>> gpurun_out/r02r.err; python3 tools/show_bench.py gpurun_out/r02r_b64.json
