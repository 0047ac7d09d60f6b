#!/bin/bash
# lane refill (several instances per lane, taken from the launch's counter) on the heavy-tailed workloads
set -o pipefail
mkdir -p gpurun_out
run() { # name ipl args...
  local name=$1 ipl=$2; shift 2
  MPC_INSTANCES_PER_LANE=$ipl python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02e_$name.json 2>> gpurun_out/r02e.err; echo "$name ipl=$ipl rc=$?"
  python tools/show_bench.py gpurun_out/r02e_$name.json | head -2
}
for ipl in 1 2 4 8; do
  run f32sweep_b262144_ipl$ipl $ipl --precision f32 --weights-sweep --no-traj --batch 262144 --inflight 1 --steps 6 --warmup 1
done
for ipl in 2 4; do
  run f32sweep_b131072_f2_ipl$ipl $ipl --precision f32 --weights-sweep --no-traj --batch 131072 --inflight 2 --steps 10 --warmup 2
done
for ipl in 1 4 8; do
  run f64sweep_b262144_ipl$ipl $ipl --weights-sweep --no-traj --batch 262144 --inflight 1 --steps 6 --warmup 1
done
for ipl in 1 4; do
  run n25_b131072_ipl$ipl $ipl --N 25 --dt 0.05 --config config-stable.json --batch 131072 --inflight 1 --steps 4 --warmup 1
done
