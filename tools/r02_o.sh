#!/bin/bash
# round 2, run o: heavy-tailed workloads -- more hardware queues / batches in flight, with and without a cut schedule
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "multi_phase" > gpurun_out/r02o_pytest.log 2>&1; echo "pytest multi-phase rc=$?"; tail -3 gpurun_out/r02o_pytest.log
b() { local name=$1; shift; python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02o_$name.json 2>> gpurun_out/r02o.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r02o_$name.json | head -1 | cut -c1-150; }
for q in 8 16 24; do
  export GPU_MAX_HW_QUEUES=$q
  for cuts in "" 20 16,16,32; do
    tag=${cuts//,/_}; tag=${tag:-none}; arg=(); [ -n "$cuts" ] && arg=(--pass-cuts "$cuts")
    b sweep32_q${q}_c${tag} --precision f32 --weights-sweep --no-traj --inflight $q --steps 96 --warmup 24 "${arg[@]}"
  done
done
export GPU_MAX_HW_QUEUES=16
b sweep32_b131072_q16_f8 --precision f32 --weights-sweep --no-traj --batch 131072 --inflight 8 --steps 48 --warmup 16
b sweep32_b131072_q16_f8_c20 --precision f32 --weights-sweep --no-traj --batch 131072 --inflight 8 --steps 48 --warmup 16 --pass-cuts 20
b sweep64_q16_cnone --weights-sweep --no-traj --inflight 16 --steps 96 --warmup 24
b sweep64_q16_c20 --weights-sweep --no-traj --inflight 16 --steps 96 --warmup 24 --pass-cuts 20
b n25_q16_cnone --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 16 --steps 96 --warmup 24
b n25_q16_c16 --N 25 --dt 0.05 --config config-stable.json --batch 32768 --inflight 16 --steps 96 --warmup 24 --pass-cuts 16
