#!/bin/bash
# round 2, second GPU pass: parity tests incl. fp32, bench lines for configs[2], [3]-per-GPU, [4]-per-GPU
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02b_pytest.log
python bench.py --steps 60 > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; echo "bench rc=$?"
python tools/show_bench.py gpurun_out/r02b_bench.json
for occ in 2 1; do
  for fl in 1 2 4; do
    MPC_F32_OCC=$occ python bench.py --precision f32 --weights-sweep --no-traj --batch 131072 --inflight $fl --steps 30 --no-cpu-baseline --no-host-leg > gpurun_out/r02b_f32_occ${occ}_f$fl.json 2>> gpurun_out/r02b_bench.err; echo "f32 occ=$occ inflight=$fl rc=$?"
    python tools/show_bench.py gpurun_out/r02b_f32_occ${occ}_f$fl.json
  done
done
python bench.py --precision f32 --steps 60 --no-cpu-baseline --no-host-leg > gpurun_out/r02b_f32_headline.json 2>> gpurun_out/r02b_bench.err; echo "f32 headline workload rc=$?"
python tools/show_bench.py gpurun_out/r02b_f32_headline.json
python bench.py --weights-sweep --no-traj --batch 131072 --inflight 4 --steps 30 --no-cpu-baseline --no-host-leg > gpurun_out/r02b_f64_sweep.json 2>> gpurun_out/r02b_bench.err; echo "f64 sweep rc=$?"
python tools/show_bench.py gpurun_out/r02b_f64_sweep.json
python bench.py --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 8 --steps 30 --no-cpu-baseline --no-host-leg > gpurun_out/r02b_n25.json 2>> gpurun_out/r02b_bench.err; echo "N25 rc=$?"
python tools/show_bench.py gpurun_out/r02b_n25.json
tail -5 gpurun_out/r02b_bench.err
