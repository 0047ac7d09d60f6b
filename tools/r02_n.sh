#!/bin/bash
# round 2, run n: multi-phase solve (cut schedules) on the heavy-tailed workloads and on the headline workload, same box
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "multi_phase" > gpurun_out/r02n_pytest.log 2>&1; echo "pytest multi-phase rc=$?"; tail -3 gpurun_out/r02n_pytest.log
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02n_$name.json 2>> gpurun_out/r02n.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02n_$name.json | head -1; }
for cuts in "" 20 16,16 16,16,32 12,12,24,48; do
  tag=${cuts//,/_}; tag=${tag:-none}; arg=(); [ -n "$cuts" ] && arg=(--pass-cuts "$cuts")
  for fl in 4 8; do
    b sweep32_c${tag}_f$fl --precision f32 --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 "${arg[@]}"
    b sweep64_c${tag}_f$fl --weights-sweep --no-traj --inflight $fl --steps 64 --warmup 16 "${arg[@]}"
  done
done
for cuts in "" 16 16,16 12,12,24; do
  tag=${cuts//,/_}; tag=${tag:-none}; arg=(); [ -n "$cuts" ] && arg=(--pass-cuts "$cuts")
  for fl in 4 8; do
    b n25_c${tag}_f$fl --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight $fl --steps 64 --warmup 16 "${arg[@]}"
  done
done
for cuts in "" 12 12,12 12,12,24; do
  tag=${cuts//,/_}; tag=${tag:-none}; arg=(); [ -n "$cuts" ] && arg=(--pass-cuts "$cuts")
  for fl in 2 3 4; do
    b head_c${tag}_f$fl --inflight $fl --steps 60 "${arg[@]}"
  done
done
