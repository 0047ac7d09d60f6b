#!/bin/bash
# experiment: what would the bulk phase alone sustain (tails dropped -- results incomplete, timing only)
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02q_$name.json 2>> gpurun_out/r02q.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02q_$name.json | head -1 | cut -c1-150; }
export MPC_EXPERIMENT_DROP_TAIL=1
for cut in 12 16 20 24; do
  for fl in 2 4 8; do
    b sweep32_dropA_c${cut}_f$fl --precision f32 --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 --pass-cuts $cut
  done
done
for fl in 2 4; do b sweep64_dropA_c20_f$fl --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 --pass-cuts 20; done
for fl in 2 4 8; do b n25_dropA_c16_f$fl --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight $fl --steps 64 --warmup 16 --pass-cuts 16; done
