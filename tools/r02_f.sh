#!/bin/bash
# profiles for round 2 (fp64 headline, fp32 headline-workload, fp32 weight sweep) + batches-in-flight sweep on the heavy-tailed workloads
set -o pipefail
mkdir -p gpurun_out
SKIP_TESTS=1 bash tools/gpu_round.sh r02 > gpurun_out/r02_round.log 2>&1; echo "round f64 rc=$?"
SKIP_TESTS=1 BENCH_ARGS="--precision f32 --inflight 4" bash tools/gpu_round.sh r02f32 > gpurun_out/r02f32_round.log 2>&1; echo "round f32 rc=$?"
python tools/show_bench.py gpurun_out/r02_bench.json gpurun_out/r02f32_bench.json
for fl in 8 16; do
  python bench.py --precision f32 --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 --no-cpu-baseline --no-host-leg > gpurun_out/r02f_f32_sweep_f$fl.json 2>> gpurun_out/r02f.err; echo "f32 sweep inflight=$fl rc=$?"
  python tools/show_bench.py gpurun_out/r02f_f32_sweep_f$fl.json | head -2
  python bench.py --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 --no-cpu-baseline --no-host-leg > gpurun_out/r02f_f64_sweep_f$fl.json 2>> gpurun_out/r02f.err; echo "f64 sweep inflight=$fl rc=$?"
  python tools/show_bench.py gpurun_out/r02f_f64_sweep_f$fl.json | head -2
done
python bench.py --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 16 --steps 64 --warmup 16 --no-cpu-baseline --no-host-leg > gpurun_out/r02f_n25_f16.json 2>> gpurun_out/r02f.err; echo "N25 inflight 16 rc=$?"
python tools/show_bench.py gpurun_out/r02f_n25_f16.json | head -2
