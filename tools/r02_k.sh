#!/bin/bash
# batched hand-over policy of the persistent kernel: correctness (bitwise tests) and lane refill on large launches
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_f32.py -m gpu -q -x > gpurun_out/r02k_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02k_pytest.log
run() { local name=$1; shift; env "$@" > /dev/null 2>&1; }
b() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --no-host-leg > gpurun_out/r02k_$name.json 2>> gpurun_out/r02k.err; echo "$name rc=$?"
  python tools/show_bench.py gpurun_out/r02k_$name.json | head -1
}
b base_f2 X=1 -- --steps 60
b base_min1 MPC_REFILL_MIN=1 -- --steps 60
b base_min16 MPC_REFILL_MIN=16 MPC_REFILL_WAIT=8 -- --steps 60
for ipl in 2 4 8; do
  b ipl${ipl}_f1 MPC_INSTANCES_PER_LANE=$ipl -- --batch $((65536*ipl)) --inflight 1 --steps 12 --warmup 2
  b ipl${ipl}_f2 MPC_INSTANCES_PER_LANE=$ipl -- --batch $((65536*ipl)) --inflight 2 --steps 12 --warmup 2
done
b ipl4_f1_min16 MPC_INSTANCES_PER_LANE=4 MPC_REFILL_MIN=16 MPC_REFILL_WAIT=6 -- --batch 262144 --inflight 1 --steps 12 --warmup 2
b ipl4_f1_min4 MPC_INSTANCES_PER_LANE=4 MPC_REFILL_MIN=4 MPC_REFILL_WAIT=2 -- --batch 262144 --inflight 1 --steps 12 --warmup 2
b ipl4_f1_min1 MPC_INSTANCES_PER_LANE=4 MPC_REFILL_MIN=1 -- --batch 262144 --inflight 1 --steps 12 --warmup 2
b f32_ipl4_f2 MPC_INSTANCES_PER_LANE=4 -- --precision f32 --batch 262144 --inflight 2 --steps 12 --warmup 2
b f32_ipl4_f1 MPC_INSTANCES_PER_LANE=4 -- --precision f32 --batch 262144 --inflight 1 --steps 12 --warmup 2
