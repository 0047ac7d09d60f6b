#!/bin/bash
# long horizon with the chain capped (max_iter 30): does re-packing pay once no straggler holds the stream?
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02w_$name.json 2>> gpurun_out/r02w.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02w_$name.json | head -1 | cut -c1-120; }
for rep in 1 2; do
for cuts in "" 12 10,6 8,4,4; do
  tag=${cuts//,/_}; tag=${tag:-none}; arg=(); [ -n "$cuts" ] && arg=(--pass-cuts "$cuts")
  b n25_mi30_c${tag}_f8_r$rep --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 8 --steps 64 --warmup 16 --max-iter 30 "${arg[@]}"
  b sweep64_mi30_c${tag}_f8_r$rep --weights-sweep --no-traj --inflight 8 --steps 64 --warmup 16 --max-iter 30 "${arg[@]}"
done
done
