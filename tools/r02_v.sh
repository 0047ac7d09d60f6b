#!/bin/bash
# long horizon: fewer, longer-lived waves (MPC_INSTANCES_PER_LANE) x batches in flight -- does a smaller live set pay?
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02v_$name.json 2>> gpurun_out/r02v.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02v_$name.json | head -1 | cut -c1-120; }
for rep in 1 2; do
for ipl in 1 8 16 32; do
  export MPC_INSTANCES_PER_LANE=$ipl
  for fl in 6 8; do
    b n25_ipl${ipl}_f${fl}_r$rep --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight $fl --steps 64 --warmup 16
  done
done
done
