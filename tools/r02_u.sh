#!/bin/bash
# traffic of the long-horizon workload (configs[3] per GPU): FETCH_SIZE / WRITE_SIZE per launch, separate passes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out
A="--N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/r02u_fetch -o pmc -- python3 $R/bench.py $A > /dev/null 2> $OUT/r02u_fetch.err; echo "fetch exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r02u_write -o pmc -- python3 $R/bench.py $A > /dev/null 2> $OUT/r02u_write.err; echo "write exit=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/r02u_sq -o pmc -- python3 $R/bench.py $A > /dev/null 2> $OUT/r02u_sq.err; echo "sq exit=$?"
python3 - <<PY
import csv, collections, glob
for sub in ("fetch", "write", "sq"):
    for f in glob.glob("$OUT/r02u_%s/**/pmc_counter_collection.csv" % sub, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mpc_solve" in r["Kernel_Name"] and int(r["Grid_Size"]) == 32768:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(sub, k, "%.6g" % (sum(v) / len(v)), len(v))
PY
