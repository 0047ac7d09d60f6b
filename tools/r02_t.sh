#!/bin/bash
# long horizon (configs[3] per GPU): lanes that take several instances in turn (MPC_INSTANCES_PER_LANE), 8 batches in flight
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02t_$name.json 2>> gpurun_out/r02t.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02t_$name.json | head -1 | cut -c1-200; }
for ipl in 1 2 4 8; do
  export MPC_INSTANCES_PER_LANE=$ipl
  for fl in 4 8; do
    b n25_ipl${ipl}_f$fl --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight $fl --steps 64 --warmup 16
  done
  b n25_b131072_ipl${ipl}_f4 --N 25 --dt 0.05 --config config-stable.json --batch 131072 --inflight 4 --steps 24 --warmup 8
done
