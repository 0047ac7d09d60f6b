#!/bin/bash
# heavy-tailed workloads with an iteration cap (the reference's own answer to stragglers is IPOPT's 0.5 s max_cpu_time)
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02s_$name.json 2>> gpurun_out/r02s.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02s_$name.json | head -1 | cut -c1-260; }
for mi in 200 100 60 40 30; do
  b sweep32_mi$mi --precision f32 --weights-sweep --no-traj --inflight 8 --steps 64 --warmup 16 --max-iter $mi
  b sweep64_mi$mi --weights-sweep --no-traj --inflight 8 --steps 64 --warmup 16 --max-iter $mi
  b n25_mi$mi --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 8 --steps 64 --warmup 16 --max-iter $mi
done
