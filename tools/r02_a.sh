#!/bin/bash
# round 2, first GPU pass: parity tests, bench, and the lane-refill potential (several instances per lane in one launch)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a_pytest.log
tail -5 gpurun_out/r02a_pytest.log
python bench.py --steps 50 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc=$?"
for cfg in "1 65536 2" "1 131072 1" "2 131072 1" "4 262144 1" "8 524288 1" "4 262144 2" "2 131072 2"; do
  set -- $cfg
  echo "ipl=$1 batch=$2 inflight=$3"
  MPC_INSTANCES_PER_LANE=$1 python bench.py --batch $2 --inflight $3 --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r02a_ipl$1_b$2_f$3.json 2>> gpurun_out/r02a_bench.err
  python - <<PY
import json
r = json.load(open("gpurun_out/r02a_ipl$1_b$2_f$3.json"))
print("   value %.3g solves/s  ms/step %.3f  per65536 %.3f ms  iters %.2f max %d" % (r["value"], r["ms_per_step"], r["ms_per_step"] * 65536 / $2, r["mean_iterations"], r["max_iterations"]))
PY
done
