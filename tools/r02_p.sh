#!/bin/bash
# round 2, run p: durations of the phases of a multi-phase solve, one batch at a time (kernel trace)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "multi_phase" > gpurun_out/r02p_pytest.log 2>&1; echo "pytest multi-phase rc=$?"; tail -3 gpurun_out/r02p_pytest.log
for cuts in 0 20 16,16,32; do
  tag=${cuts//,/_}
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02p_trace_$tag -- python3 bench.py --precision f32 --weights-sweep --no-traj --inflight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --pass-cuts $cuts > gpurun_out/r02p_$tag.json 2> gpurun_out/r02p_$tag.err
  echo "cuts $cuts rc=$?"
  python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for f in glob.glob("gpurun_out/r02p_trace_%s/**/*kernel_trace.csv" % tag, recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "mpc_solve_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows[-8:]:
        print("   start %9.3f ms  dur %8.3f ms  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Grid_Size"]))
PY
done
