#!/bin/bash
# tile pool (MPC_TILE_POOL=1): results bitwise against the plain path under concurrency, then A/B timing on the same box
mkdir -p gpurun_out
timeout -k 10 300 python tools/pool_check.py 2>&1 | tail -6
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02x_$name.json 2>> gpurun_out/r02x.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02x_$name.json | head -1 | cut -c1-120; }
for rep in 1 2; do
  for pool in 0 1; do
    export MPC_TILE_POOL=$pool
    for fl in 2 3 4; do b head_pool${pool}_f${fl}_r$rep --inflight $fl --steps 80; done
    b f32_pool${pool}_f4_r$rep --precision f32 --inflight 4 --steps 80
    b f32_pool${pool}_f6_r$rep --precision f32 --inflight 6 --steps 80
  done
done
