#!/bin/bash
# round 2, third GPU pass: fp32 tests; fp32 / fp64 headline workload against the number of batches in flight
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_f32.py -m gpu -q > gpurun_out/r02c_pytest.log 2>&1; echo "pytest f32 rc=$?"
tail -12 gpurun_out/r02c_pytest.log
for occ in 2 1; do for fl in 2 3 4 6; do
  MPC_F32_OCC=$occ python bench.py --precision f32 --inflight $fl --steps 60 --no-cpu-baseline --no-host-leg > gpurun_out/r02c_f32_occ${occ}_f$fl.json 2>> gpurun_out/r02c.err; echo "f32 occ=$occ inflight=$fl rc=$?"
  python tools/show_bench.py gpurun_out/r02c_f32_occ${occ}_f$fl.json | head -2
done; done
for fl in 2 3 4; do
  python bench.py --inflight $fl --steps 60 --no-cpu-baseline --no-host-leg > gpurun_out/r02c_f64_f$fl.json 2>> gpurun_out/r02c.err; echo "f64 inflight=$fl rc=$?"
  python tools/show_bench.py gpurun_out/r02c_f64_f$fl.json | head -2
done
for fl in 2 4; do
  python bench.py --precision f32 --weights-sweep --no-traj --inflight $fl --steps 40 --no-cpu-baseline --no-host-leg > gpurun_out/r02c_f32_sweep_f$fl.json 2>> gpurun_out/r02c.err; echo "f32 sweep 65536 inflight=$fl rc=$?"
  python tools/show_bench.py gpurun_out/r02c_f32_sweep_f$fl.json | head -2
done
